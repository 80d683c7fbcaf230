#!/bin/bash
# Register / LDS / spill figures of every kernel in a built fat object (no GPU needed):
#   profiles/tools/kernel_resources.sh historymatching_amd/csrc/press128s.o [name filter]
set -e
obj=$(readlink -f "$1")
filter=${2:-.}
tmp=$(mktemp -d)
cp "$obj" "$tmp/o.o"
(cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o >/dev/null)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp"/*gfx950* |
    grep -E "\.name:|\.vgpr_count|\.sgpr_spill|\.vgpr_spill|\.private_segment_fixed|\.group_segment_fixed|\.agpr_count" |
    awk '/agpr_count/ { if (line) print line; line = "" } { gsub(/^ +- ?/, ""); gsub(/ +/, " "); line = line " " $0 } END { print line }' |
    grep -E "$filter" || true
rm -r "$tmp"
