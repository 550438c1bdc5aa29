#!/bin/bash
bash tools/ab.sh "autotuned"
for c in 42 43 46 47 48 49 51 52 53 54 55; do bash tools/ab.sh "force-$c" YN_PW_FORCE_CFG=$c; done
bash tools/ab.sh "autotuned"
