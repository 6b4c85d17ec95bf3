mkdir -p gpurun_out/r03a
timeout 600 python -m pytest tests/test_builder_gpu.py -x -q -m gpu > gpurun_out/r03a/builder.txt 2>&1; echo "builder rc=$?" >> gpurun_out/r03a/builder.txt
tail -5 gpurun_out/r03a/builder.txt
timeout 900 python tools/chunk_c3_fused.py 267 > gpurun_out/r03a/c3.json 2> gpurun_out/r03a/c3.err; echo "c3 rc=$?"
tail -c 1500 gpurun_out/r03a/c3.err; cat gpurun_out/r03a/c3.json | head -c 3000
