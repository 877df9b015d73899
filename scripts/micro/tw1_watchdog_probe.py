import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from us_potus_model_amd import Handle, sampler, synthetic
data = synthetic.small("full")
os.environ["POTUS_DEBUG_DROP_MEMBER"] = sys.argv[1]
h = Handle(data, "full", chains=2, num_warmup=10, num_samples=0, cus_per_chain=1, twin=1, seed=3)
del os.environ["POTUS_DEBUG_DROP_MEMBER"]
h.init()
t0 = time.time()
try:
    h.run(3); print("NO ERROR")
except sampler.PotusError as e:
    print("error after %.1f s:" % (time.time() - t0), str(e)[:160])
h.close()
g = Handle(data, "full", chains=2, num_warmup=10, num_samples=0, cus_per_chain=1, twin=1, seed=3)
g.init(); g.run(3); print("fresh handle ok", g.total_leapfrogs()); g.close()
