#!/bin/bash
# usage: tools/isa_dump.sh <file.hip> <mangled-kernel-prefix> [extra hipcc flags]: device ISA of one kernel -> /tmp/k.s
set -e
f=$1; k=$2; shift 2
cd /root/repo/corsair_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../include "$@" -S --cuda-device-only $f -o /tmp/all.s 2>/dev/null
n=$(grep -n "^$k" /tmp/all.s | head -1 | cut -d: -f1)
sed -n "${n},\$p" /tmp/all.s | awk '{print} /^.Lfunc_end/{exit}' > /tmp/k.s
echo "lines $(wc -l < /tmp/k.s) mfma $(grep -c v_mfma /tmp/k.s)"
grep "^$k" -A60 /tmp/all.s | grep -i "NumVgprs:\|ScratchSize\|Occupancy\|LDSByteSize" | head -5
sed -n "${n},\$p" /tmp/all.s | grep -m4 -i "; NumVgprs:\|; ScratchSize\|; Occupancy\|; LDSByteSize"
