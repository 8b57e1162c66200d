"""Static check of the device assembly (hipcc -S --cuda-device-only): s_barrier instructions that can be reached with an LDS
STORE still in flight (no `s_waitcnt lgkmcnt(0)` between the ds_write and the barrier on some path).  The compiler leaves the
wait out in places (it treats the LDS as in-order for a workgroup); on gfx950 a store issued right in front of the barrier by
one wave was seen to lose against the loads the other waves issue right behind it (group_eig_kernel, round 4).
Round 5: the same walk for GLOBAL stores (kind="vmem": global_store / buffer_store / scratch_store / global_atomic reaching an
s_barrier without `s_waitcnt vmcnt(0)`) -- the same hazard where a workgroup shares global scratch across a barrier (the
panel eigen-solver keeps its matrix in global memory).  Most kernels store results and then meet a barrier for unrelated
reasons, so this mode is meant for a list of kernels (`only`: substrings of the mangled names).
Usage: python3 profiles/scan_barrier_waits.py file.s [...]   -- prints kernel, line number of each such barrier."""
import re, sys

def scan(path, kind="lds", only=None):
    kern, body, out = None, [], []
    if kind == "lds":
        is_store = lambda ins: ins.startswith(('ds_write', 'ds_add', 'ds_max', 'ds_min', 'ds_or'))
        drained = lambda t: 'lgkmcnt(0)' in t
    else:
        is_store = lambda ins: ins.startswith(('global_store', 'buffer_store', 'scratch_store', 'global_atomic', 'buffer_atomic', 'flat_store', 'flat_atomic'))
        drained = lambda t: 'vmcnt(0)' in t
    def flush():
        if kern is None or not body: return
        if only is not None and not any(o in kern for o in only): return
        labels = {t[:-1]: i for i, (t, _) in enumerate(body) if re.match(r'^\.LBB\d+_\d+:$', t)}
        state_in = {}  # label index -> pending flag on entry (OR over predecessors)
        changed = True
        hits = set()
        while changed:
            changed = False
            pend = False
            fall = True  # previous instruction falls through
            for i, (t, ln) in enumerate(body):
                if t.endswith(':') and t[:-1] in labels:
                    pend = (pend if fall else False) or state_in.get(i, False)
                    fall = True
                    continue
                ins = t.split()[0]
                if is_store(ins):
                    pend = True
                elif ins == 's_waitcnt' and drained(t):
                    pend = False
                elif ins == 's_barrier':
                    if pend: hits.add(ln)
                elif ins.startswith('s_cbranch') or ins == 's_branch':
                    tgt = t.split()[-1]
                    if tgt in labels:
                        j = labels[tgt]
                        if pend and not state_in.get(j, False):
                            state_in[j] = True
                            changed = True
                    if ins == 's_branch':
                        fall = False
                        continue
                elif ins in ('s_endpgm', 's_setpc_b64'):
                    fall = False
                    continue
                fall = True
        for ln in sorted(hits): out.append((kern, ln))
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m:
            flush(); kern, body = m.group(1), []
            continue
        if not t or t.startswith(';') or (t.startswith('.') and not re.match(r'^\.LBB\d+_\d+:', t)): continue
        body.append((t.split(';')[0].strip(), ln))
    flush()
    return out

if __name__ == "__main__":
    for f in sys.argv[1:]:
        for kern, ln in scan(f):
            if 'rocprim' in kern: continue
            print(f.split('/')[-1], kern[:80], 'line', ln)
