import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""dev: variant builds of the library for same-box A/B runs (tools/ab_variants.sh, tools/ab_kstats.sh): tile geometries"""
from cilantro_amd import build
variants = {
 "_v1": ["CILHIP_CUBE_EDGE=9","CILHIP_TILE_THREADS=448","CILHIP_TILE_BYTES=25088","CILHIP_TILE_MAXE=2112"],
 "_v2": ["CILHIP_CUBE_EDGE=8","CILHIP_TILE_THREADS=320","CILHIP_TILE_BYTES=17920","CILHIP_TILE_MAXE=1600"],
 "_v3": ["CILHIP_CUBE_EDGE=10","CILHIP_TILE_THREADS=576","CILHIP_TILE_BYTES=32256","CILHIP_TILE_MAXE=2720"],
}
for t in sys.argv[1:]:
    print(build.build(tag=t, defines=variants[t]))
