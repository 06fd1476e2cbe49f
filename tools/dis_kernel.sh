#!/bin/bash
# Device assembly of one kernel of the library: tools/dis_kernel.sh <mangled-name-substring> [out.s]
set -e
cd "$(dirname "$0")/../smcnuts_amd"
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -S --cuda-device-only ${SMCN_DEFS} -o /tmp/dis/api.s csrc/smcn_api.hip
python3 - "$1" "${2:-/tmp/dis/kernel.s}" <<'PY'
import sys, re
s = open('/tmp/dis/api.s').read()
pat = sys.argv[1]
m = re.search(r'^(\S*' + re.escape(pat) + r'[^\s:]*):', s, re.M)
i = m.start(); j = s.index('.Lfunc_end', i)
open(sys.argv[2], 'w').write(s[i:j])
print(m.group(1), s[i:j].count('\n'), 'lines ->', sys.argv[2])
PY
