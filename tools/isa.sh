#!/bin/bash
# tools/isa.sh <kernel-name-substring> [unit = phyhip_queue.hip] : dump the gfx950 ISA of one kernel a unit instantiates to /tmp/isa_<name>.s
# and print its resource usage.  Developer aid only.
set -e
cd /root/repo/phyml_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/phyhip_all.s ${2:-phyhip_queue.hip} 2>/dev/null
sym=$(grep -E "^_Z.*$1.*:" /tmp/phyhip_all.s | head -1 | sed 's/:.*//')
awk -v s="$sym:" 'index($0,s)==1{f=1} f{print} f&&/s_endpgm/{exit}' /tmp/phyhip_all.s > /tmp/isa_$1.s
echo "$sym -> /tmp/isa_$1.s ($(wc -l < /tmp/isa_$1.s) lines)"
awk -v s="$sym" '$0 ~ "\\.name:.*"s{f=1} f&&/vgpr_count|sgpr_count|scratch|lds_size|private_segment_fixed_size|group_segment_fixed_size/{print} f&&/\.wavefront_size/{exit}' /tmp/phyhip_all.s | sort -u
