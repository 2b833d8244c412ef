# kernel-lab: variants of the two-pass SpMV on C3 (results are wrong when ablate != 0)
# usage: bash bench/pb_lab.sh "<tag> <configs.py options>" ...
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pb_lab_$tag -o c3 -- python3 $GRAFT_REPO_ROOT/bench/configs.py c3 --steps 5 --pb 1 $@ > $GRAFT_REPO_ROOT/gpurun_out/pb_lab_$tag.log 2>&1 || exit 1
done
