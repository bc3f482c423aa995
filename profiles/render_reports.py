import os, sys
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
crt = load_package()
for name in ("cornellbox.usda", "PointInstancedMedCity.usd", "sun_sky.usda"):
    img, st = crt.render_with_report(os.path.join("scenes", name), out_exr="gpurun_out/%s.exr" % name.split(".")[0])
    print(name); print(st.report())
