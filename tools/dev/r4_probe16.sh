#!/bin/bash
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "orb or processor or clip_pipeline or smoke or c5_shape" > gpurun_out/pytest_r4o.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r4o.log
bash tools/dev/orb_stats.sh 2>&1 | tail -8
MM_ORB_RANK=count bash tools/dev/orb_stats.sh 2>&1 | grep rank
