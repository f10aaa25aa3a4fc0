#!/bin/bash
# stand-alone proposal kernel (512 threads, 2 workgroups per CU, DFT tables from L2): where its time goes
for dbg in 0 1 2 4 6 7 15 32 39 47; do
  echo "== GSM_PROPOSE_DBG=$dbg"
  GSM_PROPOSE_DBG=$dbg python scripts/kbench.py --steps 32 --reps 5 2>/dev/null | grep propose
done
echo "== NT=1024"
GSM_PROPOSE_NT=1024 python scripts/kbench.py --steps 32 --reps 5 2>/dev/null | grep propose
