#!/bin/bash
for i in 1 2 3 4 5; do bash tools/ab.sh "0.5x bs128 run $i" -- --backbone 0.5x --batch 128 --steps 60; done
