import sys, __graft_entry__ as g
from concurrent.futures import ThreadPoolExecutor
g.build()
specs = [a.split(':') for a in sys.argv[1:]]
def one(sp):
    return g.build_variant(sp[0], sp[1].split(','))
with ThreadPoolExecutor(4) as ex:
    for r in ex.map(one, specs): print(r)
