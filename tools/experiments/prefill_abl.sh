#!/bin/bash
# HERE: build one library per ablation of the prompt attention kernel (cogstream_amd/abl/libcogs_<name>.so)
set -eu
cd "$(dirname "$0")/../.."
mkdir -p cogstream_amd/abl
for v in NOEXP NOMAX NOSYNC NOWAIT NOBAR NOQK NOPV "NOQK -DPF_ABL_NOPV" "NOEXP -DPF_ABL_NOMAX"; do
  [ -n "${ONLY:-}" ] && ! echo " $ONLY " | grep -q " $v " && continue
  name=$(echo $v | sed 's/ -DPF_ABL_/_/g')
  bash tools/build_alt.sh attn -DPF_ABL_$v > /dev/null
  mv cogstream_amd/libcogs_hip_alt.so cogstream_amd/abl/libcogs_$name.so
  echo built $name
done
