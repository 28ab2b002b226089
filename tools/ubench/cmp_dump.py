import numpy as np
a=np.fromfile('/tmp/new.bin',dtype=np.float32).reshape(-1,512); b=np.fromfile('/tmp/old.bin',dtype=np.float32).reshape(-1,512)
d=np.abs(a-b); bad=d>1e-4*(1+np.abs(b))
print('bad elements',bad.sum(),'of',bad.size)
rows=np.where(bad.any(1))[0]; print('bad rows',len(rows), rows[:40])
cols=np.where(bad.any(0))[0]; print('bad cols',len(cols), cols[:64], cols[-8:] if len(cols) else '')
if len(rows):
    r=rows[0]; c=np.where(bad[r])[0]; print('row',r,'bad cols',c[:32],'new',a[r,c[:8]],'old',b[r,c[:8]])
    print('rows mod 16 histogram',np.bincount(rows%16,minlength=16))
