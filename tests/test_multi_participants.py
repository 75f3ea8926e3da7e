"""The loop over cameras on several participants from one process (rt_render_images_multi): the stripe plan on the host, and -
on one GPU, every participant a clone of the scene on that GPU - the frames of the single-scene loop bit for bit."""
import numpy as np
import pytest

from raytracer_amd import api, scenes


@pytest.mark.parametrize("height", [8, 64, 101, 1080])
@pytest.mark.parametrize("participants", [1, 2, 3, 8])
def test_stripe_plan_covers_every_row_once_and_follows_the_region_rule(height, participants):
    owner_of_memory_row = np.full(height, -1)
    for p in range(participants):
        runs = api.stripe_plan(height, participants, p)
        for first, rows in runs:
            assert 1 <= rows <= 8 and 0 <= first and first + rows <= height
            assert (owner_of_memory_row[first:first + rows] == -1).all()
            owner_of_memory_row[first:first + rows] = p
    assert (owner_of_memory_row >= 0).all()
    for y in range(height):  # image row y (0 = bottom) is memory row height - 1 - y, stripe y // 8 belongs to (y // 8) % N: rt_region's rule
        assert owner_of_memory_row[height - 1 - y] == (y // 8) % participants == api.lib().rt_stripe_owner(height, y, participants)
    assert api.lib().rt_stripe_owner(height, height, participants) == -1 and api.lib().rt_stripe_plan(height, participants, participants, None, None, 0) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("participants,frames,size", [(3, 7, (150, 101)), (2, 5, (96, 64)), (4, 3, (64, 40))])
def test_participants_on_one_gpu_render_the_frames_of_the_single_loop(participants, frames, size):
    """Clones of one scene on the same GPU stand for the GPUs of a node: stripes rendered apart, copied to the frame's root,
    bloomed there - frames, ray counts and pixel counts equal those of rt_render_images, with fewer frames than participants
    (some complete nothing) and with a height that is no multiple of the stripe."""
    sc = scenes.preset("monkey")
    g = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
    W, H = size
    feat = api.Features(enableShading=1, enableShadows=1, enableReflections=1, enableNormalInterp=1, enableAccelStructure=1, shadingModel=1,
                        enableGlossyReflection=1, numGlossySamples=3, enableBloomEffect=1)
    cams = [api.camera_from_trackball(W, H, dist=2.0 + 0.2 * k, rot_deg=(20.0, 20.0 + 50.0 * k, 0.0)) for k in range(frames)]
    want = g.render_images(cams, feat, W, H, start_depth=3)
    team = [g] + [g.clone() for _ in range(participants - 1)]
    assert all(t.device() == g.device() for t in team)
    got = api.render_images_multi(team, cams, feat, W, H, start_depth=3)
    assert np.array_equal(got["rgb"].view(np.uint32), want["rgb"].view(np.uint32))
    gs, ws = got["stats"], want["stats"]
    assert (gs.primaryRays, gs.shadowRays, gs.secondaryRays, gs.pixels) == (ws.primaryRays, ws.shadowRays, ws.secondaryRays, ws.pixels)
    again = api.render_images_multi(team, cams, feat, W, H, start_depth=3)  # the participants' buffers and streams are reused
    assert np.array_equal(again["rgb"].view(np.uint32), want["rgb"].view(np.uint32))
    with pytest.raises(api.RtError):
        api.render_images_multi([g, g], cams, feat, W, H, start_depth=3)  # one scene object twice


@pytest.mark.gpu
def test_a_cloned_scene_keeps_its_textures():
    sc = scenes.preset("teapot")
    tex = {0: np.random.RandomState(3).uniform(0, 1, (5, 7, 3)).astype(np.float32)}
    g = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"], textures=tex)
    c = g.clone()
    W, H = 96, 64
    feat = api.Features(enableShading=1, enableShadows=1, enableNormalInterp=1, enableAccelStructure=1, shadingModel=1, enableTextureMapping=1)
    cam = api.camera_from_trackball(W, H, dist=2.0)
    a, b = g.render_image(cam, feat, W, H), c.render_image(cam, feat, W, H)
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)) and np.array_equal(a["prim"], b["prim"])
    del g  # the copy owns its arrays
    b2 = c.render_image(cam, feat, W, H)
    assert np.array_equal(b2["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
