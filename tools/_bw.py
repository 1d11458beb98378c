import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for mb in (205, 822, 1644):
    a=torch.empty(mb*1000*1000//4, device='cuda'); b=torch.empty_like(a)
    ms=t(lambda: a.fill_(1.0)); print('fill %d MB: %.3f ms  %.2f TB/s write'%(mb, ms, mb/ms/1e3))
    ms=t(lambda: b.copy_(a)); print('copy %d MB: %.3f ms  %.2f TB/s r+w'%(mb, ms, 2*mb/ms/1e3))
    ms=t(lambda: a.sum()); print('sum  %d MB: %.3f ms  %.2f TB/s read'%(mb, ms, mb/ms/1e3))
    a2=torch.empty_like(a)
    def two(): a.fill_(1.0); a2.fill_(2.0)
    ms=t(two); print('2 fills %d MB each: %.3f ms  %.2f TB/s write'%(mb, ms, 2*mb/ms/1e3))
    del a,b,a2
