bash profiles/device_run.sh r05 run43 bench:1
KM_WARP_BWD_SCAN=0 bash profiles/device_run.sh r05 run43_noscan benchlite:1
KM_STAGE_TIMEOUT=150 bash profiles/device_run.sh r05 run43_suite suite
tail -5 gpurun_out/r05/run43.txt; tail -4 gpurun_out/r05/run43_noscan.txt; tail -12 gpurun_out/r05/run43_suite.txt
