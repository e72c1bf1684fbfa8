"""How many pixels / tiles / 32x8 groups take estimate_variance's short-history branch per frame (st_debug_variance_flags), 1080p Image."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from strolle_amd import Engine, scenes, CameraMode
for scene in ("cornell", "dungeon"):
    e = Engine(device=0)
    (scenes.build_cornell if scene == "cornell" else scenes.build_dungeon)(e)
    desc = (scenes.cornell_camera if scene == "cornell" else scenes.dungeon_camera)((1920, 1080), CameraMode.IMAGE)
    cam = e.create_camera(desc)
    out = torch.zeros((1080, 1920, 4), device="cuda")
    for f in range(40):
        e.update_camera(cam, desc); e.tick(0); e.render_camera(cam, out.data_ptr(), 0)
        if f in (3, 10, 20, 39):
            mask = e.variance_flags(cam); pend, groups = 0, 8100
            tiles_x = 240
            m = mask.reshape(135, 240)
            flagged_tiles = (m != 0).sum(); flagged_px = sum(bin(int(x)).count("1") for x in mask[mask != 0])
            g = (m.reshape(135, 60, 4) != 0).any(2).sum()
            print(scene, "frame", f, "pending", pend, "groups", groups, "flagged tiles", int(flagged_tiles), "of", mask.size, "flagged groups", int(g), "flagged pixels", flagged_px)
    e.close()
