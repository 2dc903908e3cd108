import sqlite3, collections, glob, sys
for d in sorted(glob.glob('gpurun_out/vpmc_*/p_results.db')):
    c=sqlite3.connect(d)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    sym=[t for t in tabs if 'kernel_symbol' in t][0]; disp=[t for t in tabs if 'kernel_dispatch' in t][0]
    pmc=[t for t in tabs if 'pmc_event' in t][0]; info=[t for t in tabs if 'info_pmc' in t][0]
    q=f"""select s.kernel_name, p.name, avg(e.value), avg(d.end-d.start) from {pmc} e join {info} p on e.pmc_id=p.id
          join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id
          where s.kernel_name like '%t256w%' or s.kernel_name like 'Custom%' group by s.kernel_name, p.name"""
    by=collections.defaultdict(dict)
    for k,n,v,dur in c.execute(q): by['vendor' if k.startswith('Custom') else 'ours'][n]=(v,dur)
    for k,dd in by.items():
        if k=='vendor' and len(sys.argv)<2: continue
        g=dd['GRBM_GUI_ACTIVE']
        print(f"{d.split('/')[1]:24s} {k:7s} {g[1]/1000:7.0f} us  clock {g[0]/g[1]:.3f} GHz  gui {g[0]:.3e} wave {dd['SQ_WAVE_CYCLES'][0]:.3e} wait_any {dd['SQ_WAIT_ANY'][0]:.3e} wait_inst {dd['SQ_WAIT_INST_ANY'][0]:.3e} active {dd['SQ_ACTIVE_INST_ANY'][0]:.3e} wait_lds {dd['SQ_WAIT_INST_LDS'][0]:.3e}")
