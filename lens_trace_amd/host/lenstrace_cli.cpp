// lenstrace_cli.cpp -- `LensTraceHIP <scene.json> [--dry-run]`: the reference's command-line driver (src/main.cpp:4-36)
// for the MI355X backend: scene file -> Camera / Model / AccelerationStructureExplicit -> RendererHIP::render -> image file.
//
// The scene schema is the reference's (src/scene_parser.cpp:3-83, defaults include/lens_trace/scene_parser.h:28-57):
//   renderer { render_platform, kernel_file_path, kernel_mode, thread_organization_mode, block_size, image_dimensions }
//   camera   { position[3], pitch, yaw, roll }        world { <name>: { file_path } }  (first model only, scene_parser.cpp:106)
//   output   { file_path }
// with "RENDER_PLATFORM_HIP" as the platform (OPENCL / CUDA scene files are accepted too: their kernel_file_path selects the
// built-in program by basename) and one optional extension object for the device-side progressive loop:
//   hip { frame_first, frame_count, accumulate, gi_max_depth, device, portable_math, strict_math, bvh: "median" | "sah" }
// Output: .jpg (the reference's format: ImageWriter, value*255 narrowed to 8 bits, quality 100; image_writer.cpp holds the
// encoder), .pfm (float RGB, bottom-up as the format demands), .ppm (8-bit, same narrowing), .raw (the float buffer as is).
// The JSON reader below is a ~100-line recursive-descent parser written for this file (objects, arrays, strings, numbers,
// true/false/null).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "lens_trace/hip/renderer_hip.h"

namespace {

struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  double number = 0;
  bool boolean = false;
  std::string string;
  std::vector<Json> array;
  std::vector<std::pair<std::string, Json>> object;   // insertion order matters: "first model only"
  const Json* find(const std::string& k) const {
    for (auto& kv : object) if (kv.first == k) return &kv.second;
    return nullptr;
  }
};

struct Parser {
  const std::string& s;
  size_t i = 0;
  std::string error;
  explicit Parser(const std::string& text) : s(text) {}
  void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
  bool fail(const char* m) { if (error.empty()) error = std::string(m) + " at offset " + std::to_string(i); return false; }
  bool parseString(std::string& out) {
    if (s[i] != '"') return fail("expected string");
    i++;
    while (i < s.size() && s[i] != '"') {
      if (s[i] == '\\' && i + 1 < s.size()) {
        char c = s[i + 1];
        out += c == 'n' ? '\n' : c == 't' ? '\t' : c;
        i += 2;
      } else out += s[i++];
    }
    if (i >= s.size()) return fail("unterminated string");
    i++;
    return true;
  }
  bool parse(Json& v) {
    ws();
    if (i >= s.size()) return fail("unexpected end");
    char c = s[i];
    if (c == '{') {
      v.kind = Json::Object; i++; ws();
      if (i < s.size() && s[i] == '}') { i++; return true; }
      for (;;) {
        ws();
        std::string key;
        if (!parseString(key)) return false;
        ws();
        if (i >= s.size() || s[i] != ':') return fail("expected ':'");
        i++;
        Json child;
        if (!parse(child)) return false;
        v.object.emplace_back(key, std::move(child));
        ws();
        if (i < s.size() && s[i] == ',') { i++; continue; }
        if (i < s.size() && s[i] == '}') { i++; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      v.kind = Json::Array; i++; ws();
      if (i < s.size() && s[i] == ']') { i++; return true; }
      for (;;) {
        Json child;
        if (!parse(child)) return false;
        v.array.push_back(std::move(child));
        ws();
        if (i < s.size() && s[i] == ',') { i++; continue; }
        if (i < s.size() && s[i] == ']') { i++; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') { v.kind = Json::String; return parseString(v.string); }
    if (s.compare(i, 4, "true") == 0) { v.kind = Json::Bool; v.boolean = true; i += 4; return true; }
    if (s.compare(i, 5, "false") == 0) { v.kind = Json::Bool; v.boolean = false; i += 5; return true; }
    if (s.compare(i, 4, "null") == 0) { v.kind = Json::Null; i += 4; return true; }
    char* end = nullptr;
    v.number = strtod(s.c_str() + i, &end);
    if (end == s.c_str() + i) return fail("unexpected character");
    v.kind = Json::Number;
    i = end - s.c_str();
    return true;
  }
};

struct SceneConfig {   // defaults of the reference's SceneParser
  std::string platform = "RENDER_PLATFORM_OPENCL";
  std::string kernelFilePath = "resources/kernels/opencl/basic.cl";
  KernelMode kernelMode = KERNEL_MODE_LINEAR;
  ThreadOrganizationMode threadOrganizationMode = THREAD_ORGANIZATION_MODE_MAX_FIT;
  uint64_t blockSize[2] = {32, 32};
  uint64_t imageDimensions[3] = {2048, 2048, 3};
  float position[3] = {0, 0, 0};
  float pitch = 0, yaw = 0, roll = 0;
  std::string modelPath;
  std::string outputPath = "output.jpg";
  uint32_t frameFirst = 0, frameCount = 0, accumulate = 0;
  int giMaxDepth = 0, device = 0;
  bool sah = false;            // default: the reference's median-split builder
  bool strictMath = false;
  bool portableMath = false;   // default: the reference kernels' own math on this GPU (BackendPropertiesHIP)
};

double num(const Json* j, double d) { return j && j->kind == Json::Number ? j->number : d; }

bool loadScene(const std::string& path, SceneConfig& c, std::string& error) {
  std::ifstream in(path);
  if (!in) { error = "cannot open " + path; return false; }
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string text = ss.str();
  Parser p(text);
  Json root;
  if (!p.parse(root) || root.kind != Json::Object) { error = "JSON: " + (p.error.empty() ? std::string("not an object") : p.error); return false; }
  if (const Json* r = root.find("renderer")) {
    if (const Json* v = r->find("render_platform")) c.platform = v->string;
    if (const Json* v = r->find("kernel_file_path")) c.kernelFilePath = v->string;
    if (const Json* v = r->find("kernel_mode")) c.kernelMode = v->string == "KERNEL_MODE_TILE" ? KERNEL_MODE_TILE : KERNEL_MODE_LINEAR;
    if (const Json* v = r->find("thread_organization_mode"))
      c.threadOrganizationMode = v->string == "THREAD_ORGANIZATION_MODE_CUSTOM" ? THREAD_ORGANIZATION_MODE_CUSTOM : THREAD_ORGANIZATION_MODE_MAX_FIT;
    if (const Json* v = r->find("block_size")) for (size_t k = 0; k < 2 && k < v->array.size(); k++) c.blockSize[k] = (uint64_t)v->array[k].number;
    if (const Json* v = r->find("image_dimensions")) for (size_t k = 0; k < 3 && k < v->array.size(); k++) c.imageDimensions[k] = (uint64_t)v->array[k].number;
  }
  if (const Json* cam = root.find("camera")) {
    if (const Json* v = cam->find("position")) for (size_t k = 0; k < 3 && k < v->array.size(); k++) c.position[k] = (float)v->array[k].number;
    c.pitch = (float)num(cam->find("pitch"), 0);
    c.yaw = (float)num(cam->find("yaw"), 0);
    c.roll = (float)num(cam->find("roll"), 0);
  }
  if (const Json* w = root.find("world"))
    if (!w->object.empty())
      if (const Json* v = w->object.front().second.find("file_path")) c.modelPath = v->string;   // first model only
  if (const Json* o = root.find("output")) if (const Json* v = o->find("file_path")) c.outputPath = v->string;
  if (const Json* h = root.find("hip")) {
    c.frameFirst = (uint32_t)num(h->find("frame_first"), 0);
    c.frameCount = (uint32_t)num(h->find("frame_count"), 0);
    if (const Json* v = h->find("accumulate")) c.accumulate = v->kind == Json::Bool ? v->boolean : (v->number != 0);
    c.giMaxDepth = (int)num(h->find("gi_max_depth"), 0);
    if (const Json* v = h->find("bvh")) c.sah = v->string == "sah";
    if (const Json* v = h->find("strict_math")) c.strictMath = v->kind == Json::Bool ? v->boolean : (v->number != 0);
    if (const Json* v = h->find("portable_math")) c.portableMath = v->kind == Json::Bool ? v->boolean : (v->number != 0);
    c.device = (int)num(h->find("device"), 0);
  }
  if (c.modelPath.empty()) { error = "scene has no world.<name>.file_path"; return false; }
  if (c.platform != "RENDER_PLATFORM_HIP" && c.platform != "RENDER_PLATFORM_OPENCL" && c.platform != "RENDER_PLATFORM_CUDA") {
    error = "unknown render_platform " + c.platform;
    return false;
  }
  return true;
}

bool endsWith(const std::string& s, const char* suffix) {
  const size_t n = strlen(suffix);
  return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

bool writeImage(std::string path, const float* rgb, uint64_t W, uint64_t H, uint64_t D) {
  if (endsWith(path, ".jpg") || endsWith(path, ".jpeg")) {   // the reference's only format (src/main.cpp:30-32)
    BufferToImageProperties b = {};
    b.sType = STRUCTURE_TYPE_BUFFER_TO_IMAGE_PROPERTIES;
    b.pBuffer = (void*)rgb;
    b.bufferSize = W * H * D * sizeof(float);
    b.imageDimensions[0] = W; b.imageDimensions[1] = H; b.imageDimensions[2] = D;
    b.imageType = IMAGE_TYPE_JPEG;
    b.filename = path.c_str();
    ImageWriter::writeBufferToImage(b);
    return true;
  }
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { printf("ERROR: cannot write %s\n", path.c_str()); return false; }
  if (endsWith(path, ".pfm")) {
    fprintf(f, "PF\n%llu %llu\n-1.0\n", (unsigned long long)W, (unsigned long long)H);
    std::vector<float> row(W * 3);
    for (uint64_t y = H; y-- > 0;) {
      for (uint64_t x = 0; x < W; x++) for (int ch = 0; ch < 3; ch++) row[x * 3 + ch] = rgb[(y * W + x) * D + ch];
      fwrite(row.data(), sizeof(float), row.size(), f);
    }
  } else if (endsWith(path, ".raw")) {
    fwrite(rgb, sizeof(float), W * H * D, f);
  } else {   // .ppm: 8 bits per channel, value * 255 narrowed like the reference's writer
    fprintf(f, "P6\n%llu %llu\n255\n", (unsigned long long)W, (unsigned long long)H);
    std::vector<unsigned char> row(W * 3);
    for (uint64_t y = 0; y < H; y++) {
      for (uint64_t x = 0; x < W; x++) for (int ch = 0; ch < 3; ch++) row[x * 3 + ch] = (unsigned char)(char)(rgb[(y * W + x) * D + ch] * 255);
      fwrite(row.data(), 1, row.size(), f);
    }
  }
  fclose(f);
  return true;
}

}  // namespace

int main(int argc, const char** argv) {
  if (argc < 2) {
    printf("usage: LensTraceHIP <scene.json> [--dry-run]\n");
    return 2;
  }
  const bool dryRun = argc > 2 && strcmp(argv[2], "--dry-run") == 0;
  SceneConfig cfg;
  std::string error;
  if (!loadScene(argv[1], cfg, error)) {
    printf("ERROR: %s\n", error.c_str());
    return 1;
  }
  printf("scene: platform=%s kernel=%s mode=%s image=%llux%llux%llu camera=(%g %g %g) yaw=%g model=%s output=%s frames=%u+%u accumulate=%u\n",
         cfg.platform.c_str(), cfg.kernelFilePath.c_str(), cfg.kernelMode == KERNEL_MODE_TILE ? "TILE" : "LINEAR",
         (unsigned long long)cfg.imageDimensions[0], (unsigned long long)cfg.imageDimensions[1], (unsigned long long)cfg.imageDimensions[2],
         cfg.position[0], cfg.position[1], cfg.position[2], cfg.yaw, cfg.modelPath.c_str(), cfg.outputPath.c_str(), cfg.frameFirst,
         cfg.frameCount, cfg.accumulate);
  if (dryRun) return 0;

  // scene_parser.cpp:97-117: Camera(position, yaw) (pitch and roll are parsed but not passed on), Model(first file), BVH
  std::unique_ptr<Camera> camera(new Camera(cfg.position[0], cfg.position[1], cfg.position[2], cfg.yaw));
  std::unique_ptr<Model> model(new Model(cfg.modelPath));
  if (!model->checkError()) return 1;
  AccelerationStructureExplicitProperties asp = {};
  asp.sType = STRUCTURE_TYPE_ACCELERATION_STRUCTURE_PROPERTIES;
  asp.accelerationStructureExplicitType = cfg.sah ? ACCELERATION_STRUCTURE_TYPE_BVH_SAH : ACCELERATION_STRUCTURE_TYPE_BVH;
  asp.pModel = model.get();
  std::unique_ptr<AccelerationStructureExplicit> as(new AccelerationStructureExplicit(asp));

  const uint64_t W = cfg.imageDimensions[0], H = cfg.imageDimensions[1], D = cfg.imageDimensions[2];
  std::vector<float> output(W * H * D, 0.0f);
  RendererHIP renderer(cfg.device);
  if (!renderer.isValid()) return 1;
  RenderPropertiesHIP rp = {};
  rp.sType = STRUCTURE_TYPE_RENDER_PROPERTIES_HIP;
  rp.kernelFilePath = cfg.kernelFilePath;
  rp.kernelMode = cfg.kernelMode;
  rp.threadOrganizationMode = cfg.threadOrganizationMode;
  rp.threadOrganization.sType = STRUCTURE_TYPE_THREAD_ORGANIZATION_HIP;
  rp.threadOrganization.blockSize[0] = cfg.blockSize[0];
  rp.threadOrganization.blockSize[1] = cfg.blockSize[1];
  for (int k = 0; k < 3; k++) rp.imageDimensions[k] = cfg.imageDimensions[k];
  rp.pOutputBuffer = output.data();
  rp.outputBufferSize = output.size() * sizeof(float);
  rp.pAccelerationStructureExplicit = as.get();
  rp.pModel = model.get();
  rp.pCamera = camera.get();
  ProgressivePropertiesHIP pp = {};
  if (cfg.frameCount > 0 || cfg.giMaxDepth > 0) {
    pp.sType = STRUCTURE_TYPE_PROGRESSIVE_PROPERTIES_HIP;
    pp.frameFirst = cfg.frameFirst;
    pp.frameCount = cfg.frameCount;
    pp.accumulate = cfg.accumulate;
    pp.giMaxDepth = cfg.giMaxDepth;
    rp.pNext = &pp;
  }
  BackendPropertiesHIP bp = {};
  if (cfg.portableMath || cfg.strictMath) {
    bp.sType = STRUCTURE_TYPE_BACKEND_PROPERTIES_HIP;
    bp.portableMath = cfg.portableMath ? 1 : 0;
    bp.strictMath = cfg.strictMath ? 1 : 0;
    bp.pNext = rp.pNext;
    rp.pNext = &bp;
  }
  renderer.render(&rp);
  return writeImage(cfg.outputPath, output.data(), W, H, D) ? 0 : 1;
}
