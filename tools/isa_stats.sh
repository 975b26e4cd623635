# Static instruction statistics of the kernels: bash tools/isa_stats.sh [extra hipcc flags]
set -e
rm -rf /tmp/isa && mkdir -p /tmp/isa && cd /tmp/isa
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I/root/repo/include -I/root/repo/raytracing_amd/csrc "$@" --save-temps -c /root/repo/raytracing_amd/csrc/rt_api.hip -o /dev/null 2>/dev/null
python3 - <<'PY'
import re, collections
lines = open('/tmp/isa/rt_api-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
starts = [(i, l.split()[1].rstrip(',')) for i, l in enumerate(lines) if l.startswith('\t.type') and '@function' in l]
for n, (i, name) in enumerate(starts):
    j = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    body = lines[i:j]
    ins = [l.strip().split()[0] for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    info = {k: next((re.search(r'(\d+)', l.split(':')[1]).group(1) for l in body if l.startswith('; ' + k + ':')), '?') for k in ('NumVgprs', 'NumSgprs', 'ScratchSize', 'Occupancy')}
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    spill = c['v_readlane_b32'] + c['v_writelane_b32']
    q = sum(c[k] for k in c if k.startswith(('v_rcp_f64', 'v_rsq_f64', 'v_div_', 'v_ldexp_f64', 'v_sqrt', 'v_exp', 'v_log', 'v_sin', 'v_cos', 'v_rcp', 'v_rsq')))
    print(f"{name[:48]:48s} instr {sum(c.values()):5d} valu {valu:5d} lanespill {spill:4d} slowops {q:4d} nop {c['s_nop']:4d}  {info}")
PY
