import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
here = os.path.dirname(os.path.abspath(__file__))
scene = ra.SceneArrays.load(os.path.join(here, "..", "tests", "golden", "scenes", "dragon.npz")).copy(width=3840, height=2160, spp=16, seed=1)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t = ra.Tracer(device=0); t.load_scene(scene)
buf = torch.empty(ra.shard_bytes(scene.width, scene.height, world), dtype=torch.uint8, device="cuda:0")
for _ in range(2): t.render_shard(0, world, buf.data_ptr())
torch.cuda.synchronize(); t.close()
