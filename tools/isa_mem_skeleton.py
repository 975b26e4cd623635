"""Print the memory instructions and waits of one kernel from /tmp/isa (after tools/isa_stats.sh): python3 tools/isa_mem_skeleton.py MANGLED_PREFIX [FROM_LINE]"""
import sys
lines = open('/tmp/isa/rt_api-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[1]) and ":" in l and not l.startswith("\t")][0]
end = [i for i, l in enumerate(lines) if i > start and '.end_amdhsa_kernel' in l][0]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for i, l in enumerate(lines[start:end]):
    t = l.strip()
    if i >= lo and (t.startswith(('global_load', 'global_store', 'scratch_', 's_barrier', 'buffer_', 'ds_read', 'ds_write')) or 'Loop Header: Depth=1' in t or 'vmcnt' in t):
        print(i, t[:100])
