cd /root/repo
for ch in 8 16 32; do echo "== TK_LOGZ_CH=$ch"; TK_LOGZ_CH=$ch timeout 200 python tools/microbench.py --reps 40 --shapes cfg2,cfg5 --ops logz 2>&1 | grep "^logz"; done
