#!/bin/bash
out=gpurun_out/r03_q; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python tools/host_profile.py 1 > $out/host_profile_p1.txt 2>&1; head -60 $out/host_profile_p1.txt
