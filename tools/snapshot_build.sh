#!/bin/bash
# copy the current working tree (sources + built .so) into _v_<name>/ so that several builds can be benchmarked on the
# SAME GPU box in one gpurun call (box-to-box variation is +-5 %, larger than most kernel changes)
set -e
cd "$(dirname "$0")/.."
d="_v_$1"; rm -rf "$d"; mkdir "$d"
cp -r bench.py udifftext_amd oracle include __graft_entry__.py BASELINE.json BASELINE.md profiles "$d"/
echo "$d"
