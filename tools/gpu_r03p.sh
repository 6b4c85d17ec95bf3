# round 3, last: the whole GPU suite and the smoke test on the round's last library
mkdir -p gpurun_out/r03p
timeout 700 python -u -m pytest tests -x -q -m gpu > gpurun_out/r03p/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03p/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03p/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03p/smoke.txt
