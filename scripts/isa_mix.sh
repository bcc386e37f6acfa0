#!/bin/bash
# usage: scripts/isa_mix.sh <unit> <kernel-name-substring>   -- instruction mix of one kernel in the gfx950 ISA
u=$1; k=$2
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -S --cuda-device-only /root/repo/hsimae_amd/csrc/$u.hip -o /tmp/$u.s 2>/dev/null
awk -v k="$k" '$0 ~ "^_ZN.*"k".*:" {f=1} f&&/s_endpgm/{f=0} f' /tmp/$u.s > /tmp/isa_k.s
echo "lines $(wc -l < /tmp/isa_k.s)"
for i in flat_load global_load global_store global_atomic ds_read_b128 ds_read_b64 ds_read_b64_tr ds_read_u16 ds_write_b16 ds_write_b64 ds_write_b128 v_mfma scratch_ s_barrier s_waitcnt v_exp v_rcp v_cndmask v_cvt_pk_bf16; do echo "$i $(grep -c $i /tmp/isa_k.s)"; done | paste - - - - - -
