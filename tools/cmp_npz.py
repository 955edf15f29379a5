import sys, numpy as np
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
bad=[k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
print("keys", len(a.files), "differing", len(bad), bad[:8])
for k in bad[:8]:
    x,y=a[k],b[k]
    print(k, x.dtype, np.max(np.abs(x.astype(float)-y.astype(float))))
