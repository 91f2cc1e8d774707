"""Scan a gfx950 assembly listing for LDS-DMA -> s_waitcnt vmcnt(0) -> LDS-DMA sequences with no barrier in between (serialised DMA issue; DESIGN.md finding 55)."""
import re,sys
# per kernel: count of "DMA ... vmcnt(0) ... DMA" occurrences where no barrier in between (serialized DMA issue)
for path in sys.argv[1:]:
    cur=None; last=None; res={}
    for ln in open(path):
        t=ln.strip()
        m=re.match(r'^(_Z\w+):',t)
        if m: cur=m.group(1); last=None; res[cur]=0; continue
        if cur is None: continue
        if t.startswith('global_load_lds'):
            if last=='wait': res[cur]+=1
            last='dma'
        elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
            if last=='dma': last='wait'
        elif t.startswith('s_barrier'): last=None
        elif t.startswith('s_endpgm'): cur=None
    for k,v in sorted(res.items(), key=lambda kv:-kv[1]):
        if v>=2: print(v,k[:120])
