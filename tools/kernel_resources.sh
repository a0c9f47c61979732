#!/bin/bash
# Register / scratch / code-size figures of every kernel in a built object (or the product library's objects by default):
# what to compare before and after touching a kernel whose codegen must not move.
# usage: bash tools/kernel_resources.sh [object.o ...]
LLVM=/opt/rocm/lib/llvm/bin
objs=("$@")
[ ${#objs[@]} -eq 0 ] && objs=(ataxxzero_amd/csrc/_obj/net_kernels.o ataxxzero_amd/csrc/_obj/engine.o)
for o in "${objs[@]}"; do
  t=$(mktemp -d)
  cp "$o" $t/x.o
  (cd $t && $LLVM/llvm-objdump --offloading x.o >/dev/null 2>&1)
  co=$(ls $t/x.o.*gfx950 2>/dev/null | head -1)
  [ -z "$co" ] && { echo "$o: no gfx950 code object"; continue; }
  echo "== $o"
  $LLVM/llvm-readelf --notes $co | grep -E "\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed|vgpr_spill" | paste - - - - - |
    sed -E 's/ +/ /g; s/\.private_segment_fixed_size:/scratch/; s/\.sgpr_count:/sgpr/; s/\.vgpr_count:/vgpr/; s/\.vgpr_spill_count:/spill/; s/\.name: //' |
    while read name rest; do printf "%-70s %s\n" "$(echo $name | c++filt | cut -c1-70)" "$rest"; done
  $LLVM/llvm-readelf -s $co | awk '$4=="FUNC"{printf "%8d B  %s\n", $3, $8}' | while read sz b n; do echo "$sz B $(echo $n | c++filt | cut -c1-80)"; done | sort -k3
  rm -rf $t
done
