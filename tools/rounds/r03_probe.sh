#!/bin/bash
out=gpurun_out/r03_probe; mkdir -p $out
hipcc --offload-arch=gfx950 -O2 tools/probe/tr_b16_probe.hip -o /tmp/tr_probe && /tmp/tr_probe > $out/tr_b16.txt 2>&1; head -60 $out/tr_b16.txt
