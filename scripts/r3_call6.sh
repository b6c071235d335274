#!/bin/bash
timeout -k 10 200 python scripts/bench_dgrad_s2.py 2>&1 | grep -v amdgpu.ids
