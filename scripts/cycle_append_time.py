import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from aligator_amd import synth
from aligator_amd.gar import ProximalRiccatiSolver
prob=synth.generate_lq_problem(21,np.zeros(36),256,36,12,mode="W")
s=ProximalRiccatiSolver(prob); s.backward(1e-12)
before=[s.getFeedback(t).copy() for t in (1,2,255)]
t0=time.perf_counter(); s.cycleAppend(prob.stages[5]); dt=time.perf_counter()-t0
assert np.array_equal(s.getFeedback(0),before[0]) and np.array_equal(s.getFeedback(1),before[1]) and np.array_equal(s.getFeedback(254),before[2])
assert np.array_equal(s.getFeedback(255),np.zeros_like(before[2]))
print(f"cycleAppend N=256: {dt*1e3:.3f} ms (rotation verified)")
