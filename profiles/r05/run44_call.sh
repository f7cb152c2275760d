# run 44: the scan-policy pair on another lease (benchlite both ways, default first), then the driver's command verbatim
bash profiles/device_run.sh r05 run44_default benchlite:1
KM_WARP_BWD_SCAN=0 bash profiles/device_run.sh r05 run44_noscan benchlite:1
bash profiles/device_run.sh r05 run44_default2 benchlite:1
bash profiles/device_run.sh r05 run44 bench:1
for f in run44_default run44_noscan run44_default2 run44; do echo "== $f"; grep -v "^settle" gpurun_out/r05/$f.txt | cut -c1-120; done
