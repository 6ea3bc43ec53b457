import numpy as np, time, sys, os
sys.path.insert(0, ".")
from limbo_amd import _capi, synth as O
eng=_capi.load_engine()
for n in (50, 200):
    rng=np.random.default_rng(5); X=rng.uniform(0,1,(n,6)); Y=O.hartmann6(X)[:,None]; om,_=O.obs_mean_data(Y)
    for server in (1,0):
        os.environ["GPE_SMALL_SERVER"]=str(server)
        h=_capi.Handle(eng); h.set_kernel(O.SE_ARD,np.zeros(7),0.01); h.set_data(X,om); h.compute()
        pts=[np.ascontiguousarray(p[None,:]) for p in rng.uniform(0,1,(400,6))]
        for p in pts[:50]: h.query_batch(p)
        best=1e9; lat=[]
        for i0 in (50,150,250):
            t0=time.perf_counter()
            for p in pts[i0:i0+100]: h.query_batch(p)
            best=min(best,time.perf_counter()-t0)
        us=h.server_last_us() if server else None
        # mu only (no forward substitution)
        t0=time.perf_counter()
        for p in pts[:100]: h.query_batch(p, want_var=False)
        mu_only=(time.perf_counter()-t0)/100*1e6
        print(f"n={n} server={server}: query {best/100*1e6:.1f} us; mu-only {mu_only:.1f} us; server copy/body us {us}; served {h.server_calls()}")
        h.close()
