#!/bin/bash
# bench.py's headline step under a few ROCm runtime settings (each in its own process): ms_per_step and ms_per_step_cold.
run() { echo -n "$1 | "; env $1 timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],5), round(j['ms_per_step_cold'],5))"; }
for rep in 1 2; do
for e in ${PAGK_ENV_PROBE:-"PAGK_NOP=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HSA_ENABLE_INTERRUPT=0 ROC_SYSTEM_SCOPE_SIGNAL=0"}; do run "$e"; done
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 ROC_SYSTEM_SCOPE_SIGNAL=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HSA_ENABLE_INTERRUPT=0"
done
