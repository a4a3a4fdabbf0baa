#!/bin/bash
# tools/kasm.sh <kernel-name-fragment> : rebuild libmgr.so, then write the device assembly of one kernel of lstm_cluster.hip to
# /tmp/k.s and print its register / scratch figures (works from any directory)
cd "$(dirname "$0")/.."
python __graft_entry__.py 2>&1 | grep -v "^\[mgr build\]" | tail -20
S=multimodal-gesture-recognition-with-lstms-and-ctc_amd/build/${2:-lstm_cluster}-hip-amdgcn-amd-amdhsa-gfx950.s
sym=$(grep -o "^_Z[A-Za-z0-9_]*$1[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
a=$(grep -n "^$sym:" $S | cut -d: -f1); b=$(grep -n "\.amdhsa_kernel $sym" $S | cut -d: -f1)
sed -n "${a},${b}p" $S > /tmp/k.s
grep -n "$sym.num_vgpr\|$sym.private_seg_size" $S | sed 's/.*\.set //'
wc -l /tmp/k.s
