cd /root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
bash tools/refresh_profiles.sh r03 bench 2>&1 | tail -3
