#!/bin/bash
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_t2.log 2>&1
rc=$?; tail -15 gpurun_out/r3_t2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/cent_only.py 2>&1 | tail -5
