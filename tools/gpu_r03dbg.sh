mkdir -p gpurun_out/r03dbg
export RELATE_AMD_MM_TRACE=1
timeout 30 python -u -c "
import numpy as np, time
from relate_amd import api
d=np.array([[0,0,1,2,2],[2,0,3,4,4],[0,0,0,1,1],[1,1,1,0,0],[1,1,1,0,0]],np.float32)
b=api.Builder(5,0.025,device=0)
print(b.build(d)[0], flush=True)
print(b.build(d)[0], flush=True)
time.sleep(1.2)
b.close()
print('closed', flush=True)
" > gpurun_out/r03dbg/t1.txt 2>&1; echo "rc=$?" >> gpurun_out/r03dbg/t1.txt
grep -v "inputs" gpurun_out/r03dbg/t1.txt | tail -8
unset RELATE_AMD_MM_TRACE
timeout 300 python -u -m pytest tests/test_builder_gpu.py -x -q -m gpu > gpurun_out/r03dbg/builder.txt 2>&1; echo "builder rc=$?" >> gpurun_out/r03dbg/builder.txt
tail -8 gpurun_out/r03dbg/builder.txt
timeout 400 python -u -m pytest tests/test_stage_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r03dbg/stage.txt 2>&1; echo "stage rc=$?" >> gpurun_out/r03dbg/stage.txt
tail -8 gpurun_out/r03dbg/stage.txt
