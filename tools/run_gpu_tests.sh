# pytest -m gpu with its log under gpurun_out/<dir> (default r4); a progress line per test file keeps the call alive
D=${1:-r4}; mkdir -p gpurun_out/$D
python -m pytest tests -m gpu -q ${FS_PYTEST_X:--x} -p no:cacheprovider --durations=15 > gpurun_out/$D/gputest.log 2>&1; rc=$?
echo "gpu tests rc $rc"; tail -25 gpurun_out/$D/gputest.log; exit $rc
