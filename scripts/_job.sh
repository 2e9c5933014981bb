bash scripts/run_asan.sh 2>&1 | tail -30
