import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from scipy import ndimage as ndi
from iterseg_amd import watershed as ws
d = np.load('tests/golden/ws_fm_10x48x48.npz')
mp = np.pad(d['mask_raw'], 1, constant_values=0)
for trial in range(3):
    roots, sizes = ws.connected_components(mp)
    lab, n = ndi.label(mp)
    r = roots[mp]
    print('trial', trial, 'gpu comps', len(np.unique(r)), 'scipy comps', n)
    # for each scipy comp, number of distinct gpu roots
    split = 0
    for c in range(1, n+1):
        u = np.unique(roots[lab == c])
        if len(u) > 1:
            split += 1
            zz, yy, xx = np.nonzero((lab == c) & (roots != u[0]))
            print('  comp', c, 'size', (lab==c).sum(), 'n roots', len(u), 'minor roots (first 8):', u[1:9],
                  'sizes', sizes.ravel()[u[1:9]], ' example voxel', zz[0], yy[0], xx[0])
    print('  split comps', split)
    sz_ok = all(sizes.ravel()[u] == (roots == u).sum() for u in np.unique(r)[:50])
    print('  sizes consistent with roots:', sz_ok)
