# kernel lab for the 2-D blocked SpMV on C3: ablations of tb_kernel (results are wrong when ablate != 0)
for ab in 0 1 2 3 4 8 16 32 12 44; do
  echo "ablate=$ab $*"
  timeout -k 10 200 python bench/configs.py c3 --steps 10 --ablate $ab "$@" 2>/dev/null | python -c "import sys,json; [print('  spmv_ms', json.loads(l)['spmv_ms']) for l in sys.stdin if l.startswith('{')]"
done
