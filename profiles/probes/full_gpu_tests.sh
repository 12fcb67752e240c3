# the whole -m gpu suite on the GPU box (what the driver runs at round end) + smoke
export TMPDIR=/tmp
mkdir -p gpurun_out/${1:-full}
timeout 3300 python -m pytest tests -m gpu -x -q > gpurun_out/${1:-full}/tests.log 2>&1
tail -8 gpurun_out/${1:-full}/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
