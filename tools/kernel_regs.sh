#!/bin/bash
# Registers / LDS / scratch of every kernel in ws_kernels.hip (device-only assembly, no GPU needed).
# usage: tools/kernel_regs.sh [extra hipcc flags]   -> /tmp/ws_kernels.s + a table on stdout
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S \
  -Iinclude -Iwater-sandbox_amd/csrc "$@" -o /tmp/ws_kernels.s water-sandbox_amd/csrc/ws_kernels.hip || exit 1
python3 - <<'PY'
import re
txt = open('/tmp/ws_kernels.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, flags=re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r'\.amdhsa_%s (\S+)' % k, body) or [None, '?'])[1]
    import subprocess
    short = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    print('%-58s vgpr %3s agpr_off %3s sgpr %3s lds %6s scratch %s' % (short[:58], g('next_free_vgpr'), g('accum_offset'), g('next_free_sgpr'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
PY
