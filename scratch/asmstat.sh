#!/bin/bash
# usage: asmstat.sh file.hip  -> per-kernel instruction statistics from the gfx950 assembly
set -e
f=$1; d=/tmp/asm_$(basename $f .hip); rm -rf $d; mkdir -p $d; cd $d
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast $EXTRA -c $f -o x.o -save-temps 2>/dev/null
s=$(ls *gfx950.s)
awk '/^_Z.*:/{name=$1} /v_mfma/{m[name]++} /v_exp_f32/{e[name]++} /v_accvgpr/{a[name]++} /s_cbranch/{b[name]++} /scratch_/{sc[name]++} /\.vgpr_count:/{} END{for(n in m) printf "%s mfma=%d exp=%d accvgpr=%d branches=%d scratch=%d\n", n, m[n], e[n], a[n], b[n], sc[n]}' $s | sort
grep -E "^\s+\.(name|vgpr_count|agpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size):" $s | paste - - - - - - | awk '{print $2, $4, $6, $8, $10, $12}' | grep -i "mfma\|chain\|potrf\|trsm" | head -70
