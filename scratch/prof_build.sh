#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
PCU_HIP_PROF_BUILD=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "prof\]" | tail -6
