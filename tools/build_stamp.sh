#!/bin/bash
# Diagnostic library with per-workgroup phase stamps in the gate / dgrad kernels (tools/clockprobe.py loads it by path).
set -e
cd "$(dirname "$0")/.."
python nasa-niswan_amd/build.py -DNINT_STAMP=${1:-2} --out=nasa-niswan_amd/build/libnint_stamp${1:-}.so
