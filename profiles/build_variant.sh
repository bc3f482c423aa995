#!/bin/bash
# Builds variants/<name>.so from a SNAPSHOT of the kernel sources, so the tree can be edited while it compiles
# (hipcc reads a .hip file twice, device pass then host pass):   bash profiles/build_variant.sh <name> [-DFLAG=V ...]
set -e
N=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/crt_variant_$N.XXXX)
mkdir -p $T/crust-render_amd $R/variants
cp -r $R/crust-render_amd/csrc $T/crust-render_amd/csrc
rm -rf $T/crust-render_amd/csrc/_obj*
cp -r $R/include $T/include
make -s -C $T/crust-render_amd/csrc OUT=$R/variants/$N.so EXTRA="$*" 2>&1 | grep -v "warning\|^$\|generated" || true
rm -rf $T
ls -la $R/variants/$N.so
