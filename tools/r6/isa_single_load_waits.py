"""Scan a gfx950 assembly listing (hipcc --save-temps) for s_waitcnt vmcnt(0) that follow exactly ONE load: per kernel the count of such waits, all vmcnt(0) waits, loads and lines (DESIGN.md finding 55)."""
import re,sys
for path in sys.argv[1:]:
    cur=None; stats={}
    pend=0
    for ln in open(path):
        ln=ln.strip()
        m=re.match(r'^(_Z\w+):',ln)
        if m:
            cur=m.group(1); stats[cur]={'loads':0,'waits0':0,'single':0,'lines':0}; pend=0; continue
        if cur is None: continue
        st=stats[cur]; st['lines']+=1
        if re.match(r'^(global_load|flat_load|buffer_load|scratch_load)',ln) and 'lds' not in ln: st['loads']+=1; pend+=1
        elif ln.startswith('s_waitcnt') and 'vmcnt(0)' in ln:
            st['waits0']+=1
            if pend==1: st['single']+=1
            pend=0
        elif ln.startswith('s_endpgm'): cur=None
    for k,v in sorted(stats.items(), key=lambda kv:-kv[1]['single']):
        if v['single']>=3: print(v['single'], v['waits0'], v['loads'], v['lines'], k[:110])
