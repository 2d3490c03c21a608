#!/bin/bash
# builds the product library + CLI, the emulation and (optionally: $1 = variant name) a copy under variants/ for tools/ab.py
R="$(cd "$(dirname "$0")/.." && pwd)"
make -s -C "$R/alignasm_amd/csrc" ARCH=gfx950 2>&1 | grep -E "error|warning: failed"
make -s -C "$R/tests/host_emul" 2>&1 | grep -E "error"
mkdir -p "$R/variants"
[ -n "$1" ] && cp "$R/alignasm_amd/libalignasm_amd.so" "$R/variants/$1.so"
ls -la --time-style=+%H:%M:%S "$R/alignasm_amd/libalignasm_amd.so" | awk '{print $6, $7}'
