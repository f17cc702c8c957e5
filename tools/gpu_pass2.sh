#!/bin/bash
timeout -k 10 900 python -u -m pytest tests/test_gpu_tokens.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -15
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
TOKEN_ROWS=kv timeout -k 10 200 python tools/token_bench.py --kv-heads 8 2>&1 | tail -5
TOKEN_ROWS=kv timeout -k 10 200 python tools/token_bench.py --kv-heads 8 --kv int8 2>&1 | tail -5
timeout -k 10 200 python tools/token_bench.py 2>&1 | tail -5
