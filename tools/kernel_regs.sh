#!/bin/bash
# Compact register / scratch report of one translation unit: tools/kernel_regs.sh spx_bwd_npb6 [filter-regex]
# (hipcc -Rpass-analysis=kernel-resource-usage; one line per kernel instance: name, VGPRs, scratch bytes per lane)
cd "$(dirname "$0")/../scaleprotoseg_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $SPX_EXTRA_HIPCC_FLAGS -Rpass-analysis=kernel-resource-usage -c "$1.hip" -o /tmp/$1.regs.o 2> /tmp/$1.rpass
awk '/Function Name|Name: /{n=$0; sub(/.*Name: /,"",n); sub(/ \[.*/,"",n)} / VGPRs: /{v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)} /ScratchSize/{s=$0; sub(/.*: /,"",s); sub(/ .*/,"",s); print n, "vgpr=" v, "scratch=" s}' /tmp/$1.rpass | grep -E "${2:-.}"
