cd /root/repo
for i in 1 2 3; do timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
