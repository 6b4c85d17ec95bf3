mkdir -p gpurun_out/r03c
timeout 250 python tools/exp_sections40.py > gpurun_out/r03c/s40_default.txt 2>&1; cat gpurun_out/r03c/s40_default.txt | tail -1
timeout 250 python tools/exp_sections40.py RELATE_AMD_MM_DEBUG=2 > gpurun_out/r03c/s40_nocol.txt 2>&1; cat gpurun_out/r03c/s40_nocol.txt | tail -1
timeout 250 python tools/exp_sections40.py LD_LIBRARY_PATH=$PWD/relate_amd/variants/ntcol > gpurun_out/r03c/s40_ntcol.txt 2>&1; cat gpurun_out/r03c/s40_ntcol.txt | tail -1
