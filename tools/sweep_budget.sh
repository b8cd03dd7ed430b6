#!/bin/bash
for b in ${BUDGETS:-128 256 384 512 768}; do echo "== budget $b"; RTMI_DEFER_BUDGET=$b timeout -k 10 100 python tools/quickbench.py 2>&1 | tail -5 | awk '{print $1, $10, $11, $12, $13, $14, $15}'; done
