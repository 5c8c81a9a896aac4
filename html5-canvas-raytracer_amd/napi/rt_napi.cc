// rt_napi.cc — thin N-API shim between the Node.js host (js/index.js) and the C ABI of
// include/rt_hip.h.  It does no rendering and no scene logic: the scene arrives already flattened
// (js/flatten.js) as one ArrayBuffer, and the frame goes back as a Uint8ClampedArray over the
// pinned host buffer the library filled — the object the reference's ImageData.data is
// (main.js:83), so `context.putImageData(new ImageData(data, w, h), 0, 0)` works unchanged.
//
// Exports:  init(maxDevices) -> deviceCount      render(blob, w, h, flags) -> {data, width, height, stats}
//           renderAsync(blob, w, h, flags) -> Promise of the same        shutdown()      abiVersion()
//           renderProgressive(blob, w, h, flags, bands, onBand(firstRow, nRows)) -> {data, promise}: `data` is the frame
//           being filled; onBand fires on the main thread as each row band lands in it; the promise resolves to the stats
// Every failure of the library becomes a thrown JS Error carrying rt_last_error().
//
// Build: g++ -shared -fPIC -I/usr/include/node rt_napi.cc -L../csrc -lrt_hip  (napi/Makefile; no node-gyp).

#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/rt_hip.h"

namespace {

#define NAPI_TRY(call)                                                        \
  do {                                                                        \
    if ((call) != napi_ok) {                                                  \
      napi_throw_error(env, nullptr, "N-API call failed: " #call);            \
      return nullptr;                                                         \
    }                                                                         \
  } while (0)

napi_value throw_rt(napi_env env, const char *what, int rc) {
  std::string msg = std::string(what) + " failed (" + std::to_string(rc) + "): " + rt_last_error();
  napi_throw_error(env, rc == RT_ERR_UNSUPPORTED ? "RT_ERR_UNSUPPORTED" : rc == RT_ERR_DEVICE ? "RT_ERR_DEVICE" : "RT_ERR", msg.c_str());
  return nullptr;
}

void free_pinned(napi_env, void *data, void *) { rt_free_pinned(data); }

struct args {
  void *blob = nullptr;      // the scene blob; `owned` when it is our (re-aligned / off-thread) copy
  bool owned = false;
  size_t bytes = 0;
  uint32_t w = 0, h = 0, flags = 0;
  uint8_t *out = nullptr;
  rt_stats st{};
  int rc = 0;
  std::string err;
  napi_deferred deferred = nullptr;
  napi_async_work work = nullptr;
};

bool parse(napi_env env, napi_callback_info info, args *a, bool copy_blob) {
  size_t argc = 4;
  napi_value argv[4];
  if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 3) {
    napi_throw_type_error(env, nullptr, "render(blob: ArrayBuffer, width, height[, flags])");
    return false;
  }
  bool is_ab = false;
  napi_is_arraybuffer(env, argv[0], &is_ab);
  void *data = nullptr;
  size_t len = 0;
  if (is_ab) napi_get_arraybuffer_info(env, argv[0], &data, &len);
  else {
    bool is_ta = false;
    napi_is_typedarray(env, argv[0], &is_ta);
    if (!is_ta) { napi_throw_type_error(env, nullptr, "scene blob must be an ArrayBuffer or a typed array"); return false; }
    napi_typedarray_type t; napi_value ab; size_t off;
    napi_get_typedarray_info(env, argv[0], &t, &len, &data, &ab, &off);
    if (t != napi_uint8_array && t != napi_uint8_clamped_array) { napi_throw_type_error(env, nullptr, "typed-array blob must be Uint8Array"); return false; }
  }
  if (napi_get_value_uint32(env, argv[1], &a->w) != napi_ok || napi_get_value_uint32(env, argv[2], &a->h) != napi_ok || a->w == 0 || a->h == 0) {
    napi_throw_type_error(env, nullptr, "width and height must be positive integers");
    return false;
  }
  if (argc >= 4) napi_get_value_uint32(env, argv[3], &a->flags);
  a->bytes = len;
  if (copy_blob || ((uintptr_t)data & 7u) != 0) {       // the library wants 8-byte alignment
    a->blob = aligned_alloc(16, (len + 15) & ~(size_t)15);
    if (!a->blob) { napi_throw_error(env, nullptr, "out of memory"); return false; }
    memcpy(a->blob, data, len);
    a->owned = true;
  } else a->blob = data;
  return true;
}

// `into`: the caller's own frame (render(..., {into})), handed back as `data`; else a new typed array over the pinned frame
napi_value make_result(napi_env env, args *a, napi_value into = nullptr) {
  const size_t n = (size_t)a->w * a->h * 4u;
  napi_value ab, ta = into, res, stats, v;
  if (!into) {
    NAPI_TRY(napi_create_external_arraybuffer(env, a->out, n, free_pinned, nullptr, &ab));
    a->out = nullptr;    // owned by the ArrayBuffer's finalizer from here on
    NAPI_TRY(napi_create_typedarray(env, napi_uint8_clamped_array, n, ab, 0, &ta));
  }
  NAPI_TRY(napi_create_object(env, &res));
  NAPI_TRY(napi_create_object(env, &stats));
  napi_set_named_property(env, res, "data", ta);
  napi_create_uint32(env, a->w, &v); napi_set_named_property(env, res, "width", v);
  napi_create_uint32(env, a->h, &v); napi_set_named_property(env, res, "height", v);
  napi_create_double(env, a->st.kernel_ms, &v); napi_set_named_property(env, stats, "kernel_ms", v);
  napi_create_double(env, a->st.total_ms, &v); napi_set_named_property(env, stats, "total_ms", v);
  napi_create_double(env, (double)a->st.pixels, &v); napi_set_named_property(env, stats, "pixels", v);
  napi_create_double(env, (double)a->st.rays, &v); napi_set_named_property(env, stats, "rays", v);
  napi_create_double(env, (double)a->st.shadow_rays, &v); napi_set_named_property(env, stats, "shadow_rays", v);
  napi_create_double(env, (double)a->st.sphere_tests, &v); napi_set_named_property(env, stats, "sphere_tests", v);
  napi_create_double(env, (double)a->st.exact_samples, &v); napi_set_named_property(env, stats, "exact_samples", v);
  { char rep[96]; napi_value sv; if (rt_elapsed_report(&a->st, rep, sizeof rep) > 0) { napi_create_string_utf8(env, rep, NAPI_AUTO_LENGTH, &sv); napi_set_named_property(env, stats, "report", sv); }
    napi_create_string_utf8(env, rt_build_id(), NAPI_AUTO_LENGTH, &sv); napi_set_named_property(env, stats, "build", sv); }
  napi_set_named_property(env, res, "stats", stats);
  return res;
}

napi_value Init(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  uint32_t maxdev = 0;
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  if (argc >= 1) napi_get_value_uint32(env, argv[0], &maxdev);
  int rc = rt_init((int)maxdev);
  if (rc != RT_OK) return throw_rt(env, "rt_init", rc);
  napi_value v;
  napi_create_int32(env, rt_device_count(), &v);
  return v;
}

// render(blob, width, height[, flags[, into]]).  `into`: a Uint8ClampedArray / Uint8Array of 4*width*height bytes that receives the
// frame - the reference creates its ImageData once (main.js:83) and every redraw fills it again (main.js:195-200).  A frame that an
// earlier render() returned is pinned memory the GPU stores into directly; any other array is pageable memory (slower, same bytes).
napi_value Render(napi_env env, napi_callback_info info) {
  args a;
  if (!parse(env, info, &a, false)) return nullptr;
  napi_value res = nullptr;
  {
    size_t argc = 5;
    napi_value argv[5];
    napi_valuetype t = napi_undefined;
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    if (argc >= 5) napi_typeof(env, argv[4], &t);
    if (argc >= 5 && t != napi_undefined && t != napi_null) {
      bool is_ta = false;
      napi_is_typedarray(env, argv[4], &is_ta);
      napi_typedarray_type tt = napi_int8_array; size_t len = 0, off = 0; void *data = nullptr; napi_value ab;
      if (is_ta) napi_get_typedarray_info(env, argv[4], &tt, &len, &data, &ab, &off);
      if (!is_ta || (tt != napi_uint8_array && tt != napi_uint8_clamped_array) || len != (size_t)a.w * a.h * 4u || !data) {
        if (a.owned) free(a.blob);
        napi_throw_type_error(env, nullptr, "into must be a Uint8ClampedArray / Uint8Array of 4*width*height bytes");
        return nullptr;
      }
      a.rc = rt_render(a.blob, a.bytes, a.w, a.h, (uint8_t *)data, a.flags, &a.st);
      res = (a.rc != RT_OK) ? throw_rt(env, "rt_render", a.rc) : make_result(env, &a, argv[4]);
      if (a.owned) free(a.blob);
      return res;
    }
  }
  a.out = (uint8_t *)rt_alloc_pinned((size_t)a.w * a.h * 4u);
  if (!a.out) res = throw_rt(env, "rt_alloc_pinned", RT_ERR_NOMEM);
  else {
    a.rc = rt_render(a.blob, a.bytes, a.w, a.h, a.out, a.flags, &a.st);
    if (a.rc != RT_OK) { rt_free_pinned(a.out); a.out = nullptr; res = throw_rt(env, "rt_render", a.rc); }
    else res = make_result(env, &a);
  }
  if (a.owned) free(a.blob);
  return res;
}

void exec_async(napi_env, void *p) {
  args *a = (args *)p;
  a->out = (uint8_t *)rt_alloc_pinned((size_t)a->w * a->h * 4u);
  if (!a->out) { a->rc = RT_ERR_NOMEM; a->err = rt_last_error(); return; }
  a->rc = rt_render(a->blob, a->bytes, a->w, a->h, a->out, a->flags, &a->st);
  if (a->rc != RT_OK) { a->err = rt_last_error(); rt_free_pinned(a->out); a->out = nullptr; }   // rt_last_error is per thread: read it here
}

void done_async(napi_env env, napi_status, void *p) {
  args *a = (args *)p;
  if (a->rc == RT_OK) {
    napi_value res = make_result(env, a);
    if (res) napi_resolve_deferred(env, a->deferred, res);
  } else {
    napi_value msg, err;
    std::string m = "rt_render failed (" + std::to_string(a->rc) + "): " + a->err;
    napi_create_string_utf8(env, m.c_str(), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &err);
    napi_reject_deferred(env, a->deferred, err);
  }
  napi_delete_async_work(env, a->work);
  if (a->owned) free(a->blob);
  delete a;
}

napi_value RenderAsync(napi_env env, napi_callback_info info) {
  args *a = new args();
  if (!parse(env, info, a, true)) { if (a->owned) free(a->blob); delete a; return nullptr; }
  napi_value promise, name;
  NAPI_TRY(napi_create_promise(env, &a->deferred, &promise));
  napi_create_string_utf8(env, "rt_render", NAPI_AUTO_LENGTH, &name);
  NAPI_TRY(napi_create_async_work(env, nullptr, name, exec_async, done_async, a, &a->work));
  NAPI_TRY(napi_queue_async_work(env, a->work));
  return promise;
}

// ---- progressive delivery: the frame buffer exists from the start (the page can keep a view of it), bands are announced
//      from the worker thread through a thread-safe function, the promise resolves when the frame is whole ----
struct prog_args : args {
  uint32_t bands = 4;
  napi_threadsafe_function tsfn = nullptr;
};
struct band_note { uint32_t row0, rows; };

void prog_call_js(napi_env env, napi_value js_cb, void *, void *data) {
  band_note *b = (band_note *)data;
  if (env && js_cb) {
    napi_value undef, argv[2];
    napi_get_undefined(env, &undef);
    napi_create_uint32(env, b->row0, &argv[0]);
    napi_create_uint32(env, b->rows, &argv[1]);
    napi_call_function(env, undef, js_cb, 2, argv, nullptr);
  }
  delete b;
}

void prog_on_band(void *user, uint32_t row0, uint32_t rows) {
  prog_args *a = (prog_args *)user;
  napi_call_threadsafe_function(a->tsfn, new band_note{row0, rows}, napi_tsfn_blocking);
}

void prog_exec(napi_env, void *p) {
  prog_args *a = (prog_args *)p;
  a->rc = rt_render_progressive(a->blob, a->bytes, a->w, a->h, a->out, a->bands, prog_on_band, a, a->flags, &a->st);
  if (a->rc != RT_OK) a->err = rt_last_error();
}

// The band notes travel through the thread-safe function's queue; the async work's completion callback may overtake the
// last of them.  So the work's completion only RELEASES the function, and the promise is settled in the function's
// finalizer, which runs on the main thread after the queue has drained: every onBand has fired before the promise resolves.
void prog_done(napi_env env, napi_status, void *p) {
  prog_args *a = (prog_args *)p;
  napi_delete_async_work(env, a->work);
  napi_release_threadsafe_function(a->tsfn, napi_tsfn_release);
}

void prog_finalize(napi_env env, void *data, void *) {
  prog_args *a = (prog_args *)data;
  if (a->rc == RT_OK) {
    napi_value stats, v;
    napi_create_object(env, &stats);
    napi_create_double(env, a->st.kernel_ms, &v); napi_set_named_property(env, stats, "kernel_ms", v);
    napi_create_double(env, a->st.total_ms, &v); napi_set_named_property(env, stats, "total_ms", v);
    napi_create_double(env, (double)a->st.pixels, &v); napi_set_named_property(env, stats, "pixels", v);
    { char rep[96]; napi_value sv; if (rt_elapsed_report(&a->st, rep, sizeof rep) > 0) { napi_create_string_utf8(env, rep, NAPI_AUTO_LENGTH, &sv); napi_set_named_property(env, stats, "report", sv); }
      napi_create_string_utf8(env, rt_build_id(), NAPI_AUTO_LENGTH, &sv); napi_set_named_property(env, stats, "build", sv); }
    napi_resolve_deferred(env, a->deferred, stats);
  } else {
    napi_value msg, err;
    std::string m = "rt_render_progressive failed (" + std::to_string(a->rc) + "): " + a->err;
    napi_create_string_utf8(env, m.c_str(), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &err);
    napi_reject_deferred(env, a->deferred, err);
  }
  if (a->owned) free(a->blob);
  delete a;
}

napi_value RenderProgressive(napi_env env, napi_callback_info info) {
  size_t argc = 6;
  napi_value argv[6];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  napi_valuetype t = napi_undefined;
  if (argc >= 6) napi_typeof(env, argv[5], &t);
  if (argc < 6 || t != napi_function) { napi_throw_type_error(env, nullptr, "renderProgressive(blob, width, height, flags, bands, onBand)"); return nullptr; }
  prog_args *a = new prog_args();
  if (!parse(env, info, a, true)) { if (a->owned) free(a->blob); delete a; return nullptr; }
  napi_get_value_uint32(env, argv[4], &a->bands);
  const size_t n = (size_t)a->w * a->h * 4u;
  a->out = (uint8_t *)rt_alloc_pinned(n);
  if (!a->out) { if (a->owned) free(a->blob); delete a; return throw_rt(env, "rt_alloc_pinned", RT_ERR_NOMEM); }
  napi_value ab, ta, res, promise, name;
  // from here on the pinned frame belongs to the ArrayBuffer (freed by its finalizer, whatever happens to the render)
  if (napi_create_external_arraybuffer(env, a->out, n, free_pinned, nullptr, &ab) != napi_ok) {
    rt_free_pinned(a->out); if (a->owned) free(a->blob); delete a;
    napi_throw_error(env, nullptr, "napi_create_external_arraybuffer failed");
    return nullptr;
  }
  NAPI_TRY(napi_create_typedarray(env, napi_uint8_clamped_array, n, ab, 0, &ta));
  NAPI_TRY(napi_create_promise(env, &a->deferred, &promise));
  napi_create_string_utf8(env, "rt_render_progressive", NAPI_AUTO_LENGTH, &name);
  NAPI_TRY(napi_create_threadsafe_function(env, argv[5], nullptr, name, 0, 1, a, prog_finalize, nullptr, prog_call_js, &a->tsfn));
  NAPI_TRY(napi_create_async_work(env, nullptr, name, prog_exec, prog_done, a, &a->work));
  NAPI_TRY(napi_queue_async_work(env, a->work));
  NAPI_TRY(napi_create_object(env, &res));
  napi_set_named_property(env, res, "data", ta);
  napi_set_named_property(env, res, "promise", promise);
  return res;
}

napi_value Shutdown(napi_env, napi_callback_info) { rt_shutdown(); return nullptr; }

// main.js:3 / :204-205: the build stamp and the end-of-frame report 'build #<id> (<elapsed>ms)'
napi_value BuildId(napi_env env, napi_callback_info) {
  napi_value v;
  napi_create_string_utf8(env, rt_build_id(), NAPI_AUTO_LENGTH, &v);
  return v;
}

napi_value AbiVersion(napi_env env, napi_callback_info) {
  napi_value v;
  napi_create_uint32(env, rt_abi_version(), &v);
  return v;
}

napi_value Validate(napi_env env, napi_callback_info info) {
  args a;
  size_t argc = 1; napi_value argv[1]; void *data = nullptr; size_t len = 0; bool is_ab = false;
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  if (argc < 1 || napi_is_arraybuffer(env, argv[0], &is_ab) != napi_ok || !is_ab) { napi_throw_type_error(env, nullptr, "validate(blob: ArrayBuffer)"); return nullptr; }
  napi_get_arraybuffer_info(env, argv[0], &data, &len);
  void *copy = aligned_alloc(16, (len + 15) & ~(size_t)15);
  memcpy(copy, data, len);
  const int rc = rt_scene_validate(copy, len);
  free(copy);
  if (rc != RT_OK) return throw_rt(env, "rt_scene_validate", rc);
  napi_value v; napi_get_boolean(env, true, &v);
  return v;
}

napi_value Module(napi_env env, napi_value exports) {
  const napi_property_descriptor props[] = {
      {"init", nullptr, Init, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"render", nullptr, Render, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"renderAsync", nullptr, RenderAsync, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"renderProgressive", nullptr, RenderProgressive, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"validate", nullptr, Validate, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"shutdown", nullptr, Shutdown, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"abiVersion", nullptr, AbiVersion, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"buildId", nullptr, BuildId, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
  };
  napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
  return exports;
}

}  // namespace

NAPI_MODULE(rt_napi, Module)
