#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
timeout 600 python tools/two_stream_probe.py 2>/dev/null | tee $O/two_stream.txt
