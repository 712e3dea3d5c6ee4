import sys
sys.path[:0]=['.','oracle','tests']
import helpers, oracle_py, numpy as np
import raytracer_amd as rt
rt.init(0)
sf,p,g=helpers.load_case('gal_shapes')
sc=rt.Scene(sf)
img=sc.render(p)
d=np.abs(img.astype(int)-g.astype(int)).max(axis=2)
ys,xs=np.nonzero(d>1)
print('bvh diff px',list(zip(ys,xs)), [ (img[y,x].tolist(), g[y,x].tolist()) for y,x in zip(ys,xs)])
p2=p.copy(); p2.use_bvh=0
img2=sc.render(p2)
print('nobvh vs golden', helpers.diff_stats(img2,g), 'nobvh vs bvh', helpers.diff_stats(img2,img))
