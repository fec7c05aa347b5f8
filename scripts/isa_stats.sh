#!/bin/bash
# Device ISA of libavp_hip.so's translation unit (no GPU needed) + per-function scratch statistics: which CALLED function spills
# inside its body, which only saves callee-saved registers in its prologue. $1 = regex over demangled function names.
cd "$(dirname "$0")/../automatedvaletparking_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-result -DAVP_BUILD $AVP_DEFS \
    --cuda-device-only -S -o /tmp/avp_dev.s avp_capi.hip 2>&1 | grep -E "error" -A6
python3 - "$1" <<'PY'
import re, subprocess, sys
txt = open('/tmp/avp_dev.s').read()
pat = sys.argv[1]
for f in re.split(r'\n(?=[_A-Za-z0-9]+:\s*; @)', txt):
    m = re.match(r'([_A-Za-z0-9]+):', f)
    if not m:
        continue
    dem = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    if not re.search(pat, dem):
        continue
    lines = f.split('\n')
    st = [i for i, l in enumerate(lines) if 'scratch_store' in l]
    ld = [i for i, l in enumerate(lines) if 'scratch_load' in l]
    vg = re.search(r'; NumVgprs: (\d+)', f)
    sp = re.search(r'; ScratchSize: (\d+)', f)
    body_st = [i for i in st if i > 100]
    body_ld = [i for i in ld if i < len(lines) - 200]
    print(dem[:110], '| lines', len(lines), 'scratch st/ld', len(st), len(ld), 'in body', len(body_st), len(body_ld), 'vgprs', vg and vg.group(1), 'scratch B', sp and sp.group(1))
PY
