#!/bin/bash
timeout -k 10 200 python -u tools/dev/cont_probe.py 16:16 8:8 || exit 1
PROBE_N=150 timeout -k 10 200 python -u tools/dev/cont_probe.py 64:64 || exit 1
QTTS_CANONICAL_PREFILL=0 timeout -k 10 200 python -u tools/dev/cont_probe.py 16:16
