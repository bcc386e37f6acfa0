#!/bin/bash
# opcode histogram of one kernel's gfx950 code: bash scripts/isa_hist.sh <unit> <mangled-name fragment> [min count]
u=$1; F=$2; M=${3:-12}
[ -f /tmp/isa_$u.s ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -S --cuda-device-only hsimae_amd/csrc/$u.hip -o /tmp/isa_$u.s 2>/dev/null
awk -v F="$F" -v M="$M" '/^_Z/ && index($0, F) && /: *;/ {f=1} f{ if ($1 ~ /^[vsdgb][a-z_0-9]+$/) c[$1]++; n++ } /s_endpgm/{if(f){printf "%s (%d lines): ", F, n; for(k in c) if (c[k] >= M) printf "%s=%d ", k, c[k]; print ""; exit}}' /tmp/isa_$u.s
