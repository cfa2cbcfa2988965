# VGPR / SGPR spill counts and scratch bytes per lane of every kernel in a HIP source: bash scratch/spills.sh iterseg_amd/csrc/conv_h2.hip
set -e
SRC=$(readlink -f ${1:-iterseg_amd/csrc/conv_h2.hip}); INC=$(dirname $(dirname $(dirname $SRC)))/include
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/spills.s $SRC -I$INC 2>/dev/null
python3 - <<'PY'
import re
s = open('/tmp/spills.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    g = lambda k: re.search(k + r':\s+(\d+)', m.group(2)).group(1)
    print(f"{m.group(1)[:70]:70s} vgpr {g('.vgpr_count'):>3s} spilled {g('.vgpr_spill_count'):>3s}  sgpr spilled {g('.sgpr_spill_count'):>3s}  scratch {g('.private_segment_fixed_size'):>4s} B/lane")
PY
