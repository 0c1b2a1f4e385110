for e in battleship battleship5; do echo "##### $e"; bash tools/gpu_pmc_quick.sh $e 2>/dev/null; done > gpurun_out/r03f_pmc_bs.txt
