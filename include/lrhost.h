/* lrhost.h — C ABI of the host-side scene library (liblrhost.so): parse a LuisaRender scene
 * description, flatten it into the POD tables of lr_scene.h, build the wide BVH, write images.
 *
 * Stands in for the reference's host build phase, which is C++ against LuisaCompute headers
 * that are absent from the snapshot:
 *   SceneParser::parse      src/sdl/scene_parser.cpp:401-407
 *   Scene::create           src/base/scene.cpp:201-233
 *   Pipeline::create        src/base/pipeline.cpp:44-99   (registries, uploads)
 *   Geometry::build         src/base/geometry.cpp:12-163
 *   save_image              src/util/imageio.cpp:694-726
 * Conventions: 0 = OK, negative = error (message via lrhost_last_error, thread-local);
 * nothing throws or aborts across this boundary.
 */
#ifndef LRHOST_H
#define LRHOST_H

#include "lr_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrhost_scene lrhost_scene;

#define LRHOST_OK 0
#define LRHOST_ERROR (-1)

/* `-D key=value` command-line macros (src/apps/cli.cpp:105-152) are passed as parallel arrays */
int lrhost_scene_load_file(const char *path, const char *const *macro_keys, const char *const *macro_values,
                           int macro_count, lrhost_scene **out);
int lrhost_scene_load_string(const char *source, const char *virtual_path, int is_json,
                             const char *const *macro_keys, const char *const *macro_values,
                             int macro_count, lrhost_scene **out);
/* bake instances + build the 4-wide BVH consumed by lrhip_upload_scene */
int lrhost_scene_build_accel(lrhost_scene *scene);
/* Motion blur.  The reference renders a frame as a sequence of shutter samples (Camera::shutter_samples,
 * src/base/camera.cpp:163-203): for each one the scene is moved to the sample's time (Pipeline::update pipeline.cpp:101-113,
 * Geometry::update geometry.cpp:194-216) and `spp` samples per pixel are rendered with their radiance scaled by `weight`
 * (src/base/integrator.cpp:91-95).  A static camera has ONE sample (shutter_span.x, 1, spp).
 *   lrhost_scene_set_time: re-evaluates every animated transform (src/transforms/lerp.cpp; instances, cameras, environment)
 *   and refits the BVH when it is built; the tables behind lrhost_scene_view change in place (*updated = whether anything
 *   moved: upload the scene again then). */
int lrhost_scene_set_time(lrhost_scene *scene, float time, int *updated);
int lrhost_scene_shutter_sample_count(const lrhost_scene *scene, int camera_index);
int lrhost_scene_shutter_sample(const lrhost_scene *scene, int camera_index, int sample_index, float *time, float *weight, uint32_t *spp);
int lrhost_scene_camera_count(const lrhost_scene *scene);
/* fill *out with pointers into `scene` (valid until lrhost_scene_destroy) */
int lrhost_scene_view(const lrhost_scene *scene, int camera_index, lr_scene *out);
const char *lrhost_scene_camera_file(const lrhost_scene *scene, int camera_index);
int lrhost_scene_has_lighting(const lrhost_scene *scene);
void lrhost_scene_destroy(lrhost_scene *scene);

int lrhost_save_image(const char *path, const float *rgba, uint32_t width, uint32_t height);
/* load to float RGBA (row 0 = top); caller frees with lrhost_free */
int lrhost_load_image(const char *path, float **rgba, uint32_t *width, uint32_t *height, uint32_t *channels);
void lrhost_free(void *p);

/* sizeof() of the lr_scene.h structs by name ("lr_scene", "lr_surface", ...), for FFI layout checks */
uint64_t lrhost_sizeof(const char *struct_name);

void lrhost_set_log_level(int level); /* 0 silent, 1 warnings, 2 info */
const char *lrhost_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* LRHOST_H */
