import sys
for ln in sys.stdin:
    if not ln.startswith("SWTS"): continue
    ev=[]
    for tok in ln.split()[1:]:
        i,a,b=tok.split(":"); ev.append((float(a),float(b)))
    n=len(ev); end=max(b for a,b in ev)
    # concurrency profile
    pts=sorted([(a,1) for a,b in ev]+[(b,-1) for a,b in ev])
    cur=0; mx=0
    for t,d in pts:
        cur+=d; mx=max(mx,cur)
    starts=sorted(a for a,b in ev); durs=sorted(b-a for a,b in ev)
    print("active WGs %d  kernel span %.1f us  max concurrent %d  start times p50 %.1f p90 %.1f max %.1f  durations p50 %.1f max %.1f" % (n,end,mx,starts[n//2],starts[int(n*0.9)],starts[-1],durs[n//2],durs[-1]))
