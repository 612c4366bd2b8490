#!/bin/bash
# The rankings of tools/l2_model at the headline size (n = 1e7 x 16), one JSON line per variant -> profiles/r06_l2_model.jsonl
# (CPU only; ~6 s per line on 8 cores).  usage: bash tools/l2_model_sweep.sh [n] [k]
cd "$(dirname "$0")/.."
N=${1:-10000000}; K=${2:-16}
g++ -O2 -std=c++17 -pthread tools/l2_model.cpp -o tools/l2_model || exit 1
O=profiles/r06_l2_model.jsonl; : > $O
run() { echo "# $*" >> $O; tools/l2_model --n $N --k $K "$@" >> $O; }
run                                   # the layout as built (gate: profiles/r03_uniform_pmc.txt: hit rate 0.832, 2.79e7 misses, 3.49 GB fetched)
for p in 13 14 15 17 18; do run --pbits $p; done                  # panel width
for s in 2 4 8 1048576; do run --slack $s; done                   # pacing window; 1048576 = no pacing
for j in 0.05 0.2 0.5; do run --jitter $j; done                   # drift between blocks
run --jitter 0.2 --slack 1048576                                  # drift and no pacing
run --jitter 0.2 --pbits 14; run --jitter 0.2 --pbits 18
for m in 2 8 16; do run --l2-mib $m; done                         # how close to capacity the XCD's working set is
run --warm 1 --lead 2; run --warm 4 --lead 2; run --warm 8 --lead 4      # the warmer split of DESIGN §10: warm blocks per XCD that only stream
run --w 2500000; run --w 2500000 --xcd-spans 1                    # config 5's locality-bounded columns on one GPU, spans dealt inside an L2 or not
run --w 600000; run --w 600000 --xcd-spans 1
run --cus 240; run --cus 224                                      # fewer CUs (what a split takes away from the gathering side): a third round?
# the same questions with the stream's non-temporal loads installed as their set's next victim (--nt 1; how gfx950's L2 treats them is not documented)
run --nt 1; run --nt 1 --pbits 14; run --nt 1 --pbits 17; run --nt 1 --slack 2; run --nt 1 --slack 1048576; run --nt 1 --jitter 0.05; run --nt 1 --jitter 0.2
run --nt 1 --l2-mib 2; run --nt 1 --warm 4 --lead 2; run --nt 1 --w 600000; run --nt 1 --w 600000 --xcd-spans 1; run --nt 1 --w 2500000; run --nt 1 --w 2500000 --xcd-spans 1
