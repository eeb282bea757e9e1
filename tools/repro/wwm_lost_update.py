"""Forward may-analysis on AMDGPU assembly: for every VGPR that receives SGPR spills (v_writelane_b32) AND is itself spilled / reloaded, is there a path on which lanes
written since the last store are still unsaved when the register is reloaded from its slot (the update would be lost)?"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
only = sys.argv[2] if len(sys.argv) > 2 else None
i = 0
funcs = []
cur = None
for ln in lines:
    m = re.match(r'^(_Z[\w.$]*|[A-Za-z_][\w.$]*):\s', ln + ' ')
    if m and not ln.startswith('.L'):
        cur = [m.group(1), []]; funcs.append(cur); continue
    if ln.startswith('.Lfunc_end'):
        cur = None; continue
    if cur is not None: cur[1].append(ln)
for name, body in funcs:
    if only and only not in name: continue
    # blocks
    blocks = collections.OrderedDict(); label = 'entry'; blocks[label] = []
    for ln in body:
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            label = m.group(1); blocks[label] = []; continue
        t = ln.strip()
        if not t or t.startswith(';') or t.startswith('.'): continue
        blocks[label].append(t)
    names = list(blocks)
    succ = {}
    for k, b in enumerate(names):
        s = []
        fall = True
        for t in blocks[b]:
            m = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', t)
            if m: s.append(m.group(1))
            m = re.match(r's_branch (\.LBB\d+_\d+)', t)
            if m: s.append(m.group(1)); fall = False
            if t.startswith('s_endpgm') or t.startswith('s_setpc_b64'): fall = False
        if fall and k + 1 < len(names): s.append(names[k + 1])
        succ[b] = s
    regs = set()
    for b in blocks.values():
        for t in b:
            m = re.match(r'v_writelane_b32 (v\d+),', t)
            if m: regs.add(m.group(1))
    for reg in sorted(regs):
        has_reload = any(re.match(r'scratch_load_dword %s,' % reg, t) for b in blocks.values() for t in b)
        if not has_reload: continue
        # state: the stack slots the register's contents may currently correspond to, and the lanes written since (may-analysis: unions at joins).
        # A reload from the slot the register was last stored to / loaded from, while lanes are unsaved, loses them; a reload from ANOTHER slot
        # (a function epilogue restoring its caller's value) discards them on purpose.
        IN = {b: (frozenset(), frozenset()) for b in names}
        work = collections.deque(names)
        reports = {}
        def transfer(b, st, report):
            cur, dirty = set(st[0]), set(st[1])
            for idx, t in enumerate(blocks[b]):
                m = re.match(r'v_writelane_b32 %s, (\S+), (\d+)' % reg, t)
                if m:
                    for c in (cur or {None}): dirty.add((c, int(m.group(2))))
                    continue
                m = re.match(r'scratch_store_dword off, %s, (.*?)(;|$)' % reg, t)
                if m: cur = {m.group(1).strip()}; dirty.clear(); continue
                m = re.match(r'scratch_load_dword %s, (.*?)(;|$)' % reg, t)
                if m:
                    slot = m.group(1).strip()
                    lost = sorted(l for (c, l) in dirty if c == slot)
                    if lost and report is not None: report[(b, idx)] = lost
                    cur = {slot}; dirty.clear(); continue
            return (frozenset(cur), frozenset(dirty))
        while work:
            b = work.popleft()
            out = transfer(b, IN[b], None)
            for s in succ[b]:
                if s in IN and not (out[0] <= IN[s][0] and out[1] <= IN[s][1]):
                    IN[s] = (IN[s][0] | out[0], IN[s][1] | out[1])
                    if s not in work: work.append(s)
        for b in names: transfer(b, IN[b], reports)
        print('%-50s %-5s blocks %4d  reloads with unsaved lanes on some path: %d' % (name[:50], reg, len(names), len(reports)))
        for (b, idx), lanes in list(reports.items())[:6]:
            print('      in %s at instruction %d: lanes %s' % (b, idx, lanes))

def path_without_store(fname, reg, src, dst):
    for name, body in funcs:
        if fname not in name: continue
        blocks = collections.OrderedDict(); label = 'entry'; blocks[label] = []
        for ln in body:
            m = re.match(r'^(\.LBB\d+_\d+):', ln)
            if m: label = m.group(1); blocks[label] = []; continue
            t = ln.strip()
            if not t or t.startswith(';') or t.startswith('.'): continue
            blocks[label].append(t)
        names = list(blocks); succ = {}
        for k, b in enumerate(names):
            s = []; fall = True
            for t in blocks[b]:
                m = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', t)
                if m: s.append(m.group(1))
                m = re.match(r's_branch (\.LBB\d+_\d+)', t)
                if m: s.append(m.group(1)); fall = False
                if t.startswith('s_endpgm') or t.startswith('s_setpc_b64'): fall = False
            if fall and k + 1 < len(names): s.append(names[k + 1])
            succ[b] = s
        def touches(b):
            for t in blocks[b]:
                if re.match(r'scratch_store_dword off, %s,' % reg, t): return 'store'
                if re.match(r'scratch_load_dword %s,' % reg, t): return 'load'
            return None
        prev = {src: None}; q = collections.deque([src])
        while q:
            b = q.popleft()
            for s in succ[b]:
                if s == dst:
                    p = [dst, b]
                    while prev[p[-1]] is not None: p.append(prev[p[-1]])
                    return list(reversed(p)), blocks
                if s in prev: continue
                if touches(s): continue          # a block that stores or reloads the register ends the search on that path
                prev[s] = b; q.append(s)
        return None, blocks
if len(sys.argv) > 5:
    p, blocks = path_without_store(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    print('path:', p)
    if p:
        for b in p:
            print('==', b, len(blocks[b]), 'instructions;', [t for t in blocks[b] if re.search(r'%s|s_cbranch|s_branch|s_swappc' % sys.argv[3], t)][:12])
