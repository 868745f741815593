"""The drop-in boundary is a C ABI: include/lrhost.h + include/lrhip.h must compile as plain C99, link against the two shared
libraries and behave from a C program (INTEGRATION.md §3 "Minimal C driver").  Without a GPU lrhip_create must fail with an error
code and a message (no abort, no exception across the boundary); the host half (parse, flatten, BVH, shutter samples, image IO)
runs for real."""
import os
import shutil
import subprocess

import pytest

from luisarender_amd import _ffi
from luisarender_amd.scenes import cornell_box

DRIVER = r"""
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lrhost.h"
#include "lrhip.h"

int main(int argc, char **argv) {
    lrhost_scene *s = NULL;
    lr_scene view;
    lrhip_ctx *ctx = NULL;
    if (argc < 3) { return 2; }
    lrhost_set_log_level(0);
    if (lrhost_scene_load_file(argv[1], NULL, NULL, 0, &s) != LRHOST_OK) { fprintf(stderr, "load: %s\n", lrhost_last_error()); return 3; }
    if (lrhost_scene_build_accel(s) != LRHOST_OK) { fprintf(stderr, "accel: %s\n", lrhost_last_error()); return 4; }
    if (lrhost_scene_view(s, 0, &view) != LRHOST_OK) { return 5; }
    printf("res %ux%u spp %u tris %u nodes %u shutter_samples %d\n", view.camera.width, view.camera.height, view.camera.spp,
           view.accel.triangle_count, view.accel.node_count, lrhost_scene_shutter_sample_count(s, 0));
    {
        float t = -1.f, w = -1.f;
        uint32_t n = 0;
        if (lrhost_scene_shutter_sample(s, 0, 0, &t, &w, &n) != LRHOST_OK || n != view.camera.spp || w != 1.f) { return 6; }
        if (lrhost_scene_shutter_sample(s, 0, 7, &t, &w, &n) == LRHOST_OK) { return 7; } /* out of range -> error code + message */
        if (strlen(lrhost_last_error()) == 0) { return 8; }
    }
    {
        int rc = lrhip_create(0, &ctx);
        if (rc == LRHIP_OK) {
            lrhip_render_params p;
            float *rgba = (float *)malloc(16u * (size_t)view.camera.width * view.camera.height);
            memset(&p, 0, sizeof(p));
            p.spp_end = view.camera.spp;
            p.tile_end = ((view.camera.width + 7) / 8) * ((view.camera.height + 7) / 8);
            p.tile_stride = 1;
            if (lrhip_upload_scene(ctx, &view) != LRHIP_OK || lrhip_render(ctx, &p) != LRHIP_OK ||
                lrhip_film_download(ctx, rgba, 1) != LRHIP_OK) { fprintf(stderr, "device: %s\n", lrhip_last_error()); return 9; }
            if (lrhost_save_image(argv[2], rgba, view.camera.width, view.camera.height) != LRHOST_OK) { return 10; }
            printf("rendered %s centre %.4f\n", argv[2], rgba[4 * ((view.camera.height / 2) * view.camera.width + view.camera.width / 2)]);
            free(rgba);
            lrhip_destroy(ctx);
        } else {
            printf("no device: rc %d (%s)\n", rc, lrhip_last_error());
            if (rc >= 0 || strlen(lrhip_last_error()) == 0) { return 11; }
        }
    }
    lrhost_scene_destroy(s);
    return 0;
}
"""


def _build(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    _ffi.host_lib()
    src = tmp_path / "driver.c"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    inc = os.path.join(_ffi.REPO_ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe),
                           "-L", _ffi.LIB_DIR, "-llrhost", "-llrhip", f"-Wl,-rpath,{_ffi.LIB_DIR}"])
    scene = tmp_path / "cornell.luisa"
    scene.write_text(cornell_box(resolution=32, spp=4))
    return exe, scene


def test_headers_are_c99_and_the_host_half_runs_from_c(tmp_path):
    exe, scene = _build(tmp_path)
    r = subprocess.run([str(exe), str(scene), str(tmp_path / "out.exr")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "res 32x32 spp 4 tris 3" in r.stdout and "shutter_samples 1" in r.stdout
    assert "no device: rc -" in r.stdout or "rendered" in r.stdout


@pytest.mark.gpu
def test_c_driver_renders_on_the_device(tmp_path):
    from luisarender_amd.scene import load_image
    exe, scene = _build(tmp_path)
    r = subprocess.run([str(exe), str(scene), str(tmp_path / "out.exr")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rendered" in r.stdout, (r.returncode, r.stdout, r.stderr)
    img, _ = load_image(str(tmp_path / "out.exr"))
    assert img.shape == (32, 32, 4) and img[..., :3].mean() > 0.01
