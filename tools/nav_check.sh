# Maze/Nav path: parity tests that touch the generator, then the Nav env-only profile (tools/prof_nav.sh)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_drivers_gpu.py -x -q -m gpu 2>&1 | tail -5
bash tools/prof_nav.sh
