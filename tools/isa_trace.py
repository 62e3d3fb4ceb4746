"""dev: condensed memory / wait / barrier trace of one kernel out of a hipcc -S listing:
   python tools/isa_trace.py <file.s> <mangled-name substring> [first line]"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ':') if (name + ':') in s else s.index(name)
j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
keep = re.compile(r'buffer_load|global_load|scratch_|global_store|vmcnt|s_barrier|v_mfma|ds_read_b128|ds_write_b128')
last = None; cnt = 0
for n, l in enumerate(body):
    t = l.strip()
    if n < first or not keep.search(t):
        continue
    key = t.split()[0]
    if key == last and key in ('ds_read_b128', 'ds_write_b128', 'v_mfma_f64_16x16x4_f64', 'buffer_load_dword', 'scratch_load_dword', 'scratch_store_dword'):
        cnt += 1; continue
    if cnt: print(f"      ... x{cnt} more {last}"); cnt = 0
    print(f"{n:5d} {t[:120]}"); last = key
