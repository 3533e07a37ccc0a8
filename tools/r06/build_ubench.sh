#!/bin/bash
# builds the round-6 micro-benchmarks into tools/ubench/_build (git-ignored *.out; they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../ubench"
mkdir -p _build
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -I../../include -Wall -Wno-unused-function"
for u in "$@"; do /opt/rocm/bin/hipcc $F $EXTRA -o _build/$u.out $u.hip -ldl; done
