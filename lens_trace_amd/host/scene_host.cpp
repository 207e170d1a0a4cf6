// scene_host.cpp -- host-side scene classes behind the renderer plugin: Camera, Model (own .obj/.mtl subset
// reader) and AccelerationStructureExplicit (own deterministic BVH builder).  Written from scratch; what is
// kept from the reference is the *data contract*: the byte layouts of the five buffers the plugin uploads and
// the semantics that shape them --
//   PrimitiveInfo bounds / AABB-centre "centroid"            reference src/model.cpp:37-73
//   Material {diffuse, ior, dissolve, emission}               src/model.cpp:75-81
//   median split on the largest centroid-extent axis, pre-order flattening (left child = i+1),
//   BVH-ordered primitive copy, emissive-triangle list       src/acceleration_structure_explicit.cpp:3-169
// Deliberate differences: centroid bounds are initialised (the reference reads them
// uninitialised, so its tree depends on stack garbage); a leaf always holds exactly one triangle (the
// reference's traversal only ever intersects the first triangle of a leaf); inputs are range-checked.
#include "lens_trace/hip/lens_trace_api.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <limits>
#include <map>
#include <numeric>
#include <sstream>

// ------------------------------------------------------------------------------------------ Camera
Camera::Camera(float px, float py, float pz, float yaw_, float pitch_, float roll_)
    : yaw(yaw_), pitch(pitch_), roll(roll_), frameCount(0) {
  position[0] = px; position[1] = py; position[2] = pz;
  sync();
}
Camera::~Camera() {}
void Camera::sync() {   // {pos[3], yaw, pitch, roll, frameCount bit-copied into the 7th 4-byte slot}
  memcpy(buffer + 0, position, 12);
  memcpy(buffer + 12, &yaw, 4);
  memcpy(buffer + 16, &pitch, 4);
  memcpy(buffer + 20, &roll, 4);
  memcpy(buffer + 24, &frameCount, 4);
}
float Camera::getPositionX() { return position[0]; }
float Camera::getPositionY() { return position[1]; }
float Camera::getPositionZ() { return position[2]; }
float Camera::getYaw() { return yaw; }
float Camera::getPitch() { return pitch; }
float Camera::getRoll() { return roll; }
uint32_t Camera::getFrameCount() { return frameCount; }
void Camera::setPosition(float x, float y, float z) { position[0] = x; position[1] = y; position[2] = z; sync(); }
void Camera::updatePosition(float x, float y, float z) { position[0] += x; position[1] += y; position[2] += z; sync(); }
void Camera::setRotation(float y, float p, float r) { yaw = y; pitch = p; roll = r; sync(); }
void Camera::updateRotation(float y, float p, float r) { yaw += y; pitch += p; roll += r; sync(); }
void Camera::incrementFrameCount() { frameCount += 1; sync(); }
void Camera::resetFrameCount() { frameCount = 0; sync(); }
void* Camera::getCameraBuffer() { return buffer; }
uint64_t Camera::getCameraBufferSize() { return sizeof(buffer); }

// ------------------------------------------------------------------------------------------ Model
void Model::addTriangle(const float* p, const float* n, int materialIndex) {
  PrimitiveInfo info;
  memcpy(info.positionA, p + 0, 12); memcpy(info.positionB, p + 3, 12); memcpy(info.positionC, p + 6, 12);
  memcpy(info.normalA, n + 0, 12); memcpy(info.normalB, n + 3, 12); memcpy(info.normalC, n + 6, 12);
  info.materialIndex = materialIndex;
  for (int a = 0; a < 3; a++) {
    info.boundsMin[a] = std::min(std::min(p[a], p[3 + a]), p[6 + a]);
    info.boundsMax[a] = std::max(std::max(p[a], p[3 + a]), p[6 + a]);
    info.centroid[a] = 0.5f * info.boundsMin[a] + 0.5f * info.boundsMax[a];
  }
  primitiveInfoList.push_back(info);
}

Model::Model(const float* positions, const float* normals, const int* materialIndices, uint64_t triangleCount,
             const Material* materials, uint64_t materialCount)
    : success(true) {
  fileName = "<memory>";
  materialList.assign(materials, materials + materialCount);
  primitiveInfoList.reserve(triangleCount);
  for (uint64_t t = 0; t < triangleCount; t++) {
    int m = materialIndices ? materialIndices[t] : 0;
    if (m < 0 || (uint64_t)m >= materialCount) {
      error = "material index out of range";
      success = false;
      m = 0;
    }
    addTriangle(positions + 9 * t, normals + 9 * t, m);
  }
  if (materialCount == 0) { error = "no materials"; success = false; }
}

namespace {

struct ObjIndex { int v, vt, vn; };

// one "v", "v/vt", "v//vn" or "v/vt/vn" token; indices are 1-based, negative = relative to the end
bool parseFaceToken(const std::string& tok, int nv, int nvn, ObjIndex& out) {
  int vals[3] = {0, 0, 0};
  int field = 0;
  size_t i = 0;
  while (i <= tok.size() && field < 3) {
    size_t j = tok.find('/', i);
    if (j == std::string::npos) j = tok.size();
    if (j > i) vals[field] = atoi(tok.substr(i, j - i).c_str());
    field++;
    i = j + 1;
  }
  auto fix = [](int idx, int n) { return idx > 0 ? idx - 1 : (idx < 0 ? n + idx : -1); };
  out.v = fix(vals[0], nv);
  out.vt = -1;
  out.vn = fix(vals[2], nvn);
  return out.v >= 0 && out.v < nv;
}

// Ear clipping of a planar polygon (given as indices into the position array), projected on the plane of its
// dominant normal axis.  Emits triangles as index triples into `poly`.
void earClip(const std::vector<const float*>& pts, std::vector<int>& tris) {
  const int n = (int)pts.size();
  double nx = 0, ny = 0, nz = 0;   // Newell normal
  for (int i = 0; i < n; i++) {
    const float* a = pts[i];
    const float* b = pts[(i + 1) % n];
    nx += ((double)a[1] - b[1]) * ((double)a[2] + b[2]);
    ny += ((double)a[2] - b[2]) * ((double)a[0] + b[0]);
    nz += ((double)a[0] - b[0]) * ((double)a[1] + b[1]);
  }
  int ax0 = 0, ax1 = 1;
  double sign = nz;
  if (fabs(nx) >= fabs(ny) && fabs(nx) >= fabs(nz)) { ax0 = 1; ax1 = 2; sign = nx; }
  else if (fabs(ny) >= fabs(nz)) { ax0 = 2; ax1 = 0; sign = ny; }
  const double s = sign < 0 ? -1.0 : 1.0;
  auto cross2 = [&](int a, int b, int c) {
    const double ux = (double)pts[b][ax0] - pts[a][ax0], uy = (double)pts[b][ax1] - pts[a][ax1];
    const double vx = (double)pts[c][ax0] - pts[a][ax0], vy = (double)pts[c][ax1] - pts[a][ax1];
    return s * (ux * vy - uy * vx);
  };
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  int guard = 0;
  while (idx.size() > 3 && guard < 4 * n) {
    bool clipped = false;
    const int m = (int)idx.size();
    for (int i = 0; i < m; i++) {
      const int a = idx[(i + m - 1) % m], b = idx[i], c = idx[(i + 1) % m];
      if (cross2(a, b, c) <= 0) continue;   // reflex or degenerate corner
      bool inside = false;
      for (int k = 0; k < m && !inside; k++) {
        const int p = idx[k];
        if (p == a || p == b || p == c) continue;
        inside = cross2(a, b, p) >= 0 && cross2(b, c, p) >= 0 && cross2(c, a, p) >= 0;
      }
      if (inside) continue;
      tris.push_back(a); tris.push_back(b); tris.push_back(c);
      idx.erase(idx.begin() + i);
      clipped = true;
      break;
    }
    if (!clipped) {   // numerically stuck (collinear run): drop a vertex with a fan triangle
      tris.push_back(idx[0]); tris.push_back(idx[1]); tris.push_back(idx[2]);
      idx.erase(idx.begin() + 1);
    }
    guard++;
  }
  if (idx.size() == 3) { tris.push_back(idx[0]); tris.push_back(idx[1]); tris.push_back(idx[2]); }
}

void loadMtl(const std::string& path, std::vector<Material>& mats, std::map<std::string, int>& names, std::string& warning) {
  std::ifstream in(path);
  if (!in) { warning += "material file not found: " + path + "\n"; return; }
  std::string line;
  Material* cur = nullptr;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string key;
    if (!(ss >> key) || key[0] == '#') continue;
    if (key == "newmtl") {
      std::string name;
      ss >> name;
      Material m;
      m.diffuse[0] = m.diffuse[1] = m.diffuse[2] = 0.0f;   // tinyobj defaults: Kd 0, Ni 1, d 1, Ke 0
      m.ior = 1.0f; m.dissolve = 1.0f;
      m.emission[0] = m.emission[1] = m.emission[2] = 0.0f;
      names[name] = (int)mats.size();
      mats.push_back(m);
      cur = &mats.back();
    } else if (cur) {
      if (key == "Kd") ss >> cur->diffuse[0] >> cur->diffuse[1] >> cur->diffuse[2];
      else if (key == "Ke") ss >> cur->emission[0] >> cur->emission[1] >> cur->emission[2];
      else if (key == "Ni") ss >> cur->ior;
      else if (key == "d") ss >> cur->dissolve;
      else if (key == "Tr") { float tr; if (ss >> tr) cur->dissolve = 1.0f - tr; }
    }
  }
}

}  // namespace

Model::Model(std::string fileName_) : fileName(fileName_), success(false) {
  std::ifstream in(fileName);
  if (!in) {
    error = "cannot open " + fileName;
    checkError();
    return;
  }
  const size_t slash = fileName.find_last_of('/');
  const std::string baseDir = slash == std::string::npos ? std::string() : fileName.substr(0, slash + 1);
  std::vector<float> v, vn;
  std::map<std::string, int> materialNames;
  int currentMaterial = -1;
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ss(line);
    std::string key;
    if (!(ss >> key) || key[0] == '#') continue;
    if (key == "v") {
      float x = 0, y = 0, z = 0; ss >> x >> y >> z;
      v.push_back(x); v.push_back(y); v.push_back(z);
    } else if (key == "vn") {
      float x = 0, y = 0, z = 0; ss >> x >> y >> z;
      vn.push_back(x); vn.push_back(y); vn.push_back(z);
    } else if (key == "mtllib") {
      std::string name; ss >> name;
      loadMtl(baseDir + name, materialList, materialNames, warning);
    } else if (key == "usemtl") {
      std::string name; ss >> name;
      auto it = materialNames.find(name);
      if (it == materialNames.end()) { warning += "unknown material " + name + "\n"; currentMaterial = -1; }
      else currentMaterial = it->second;
    } else if (key == "f") {
      std::vector<ObjIndex> face;
      std::string tok;
      bool ok = true;
      while (ss >> tok) {
        ObjIndex idx;
        if (!parseFaceToken(tok, (int)v.size() / 3, (int)vn.size() / 3, idx)) { ok = false; break; }
        face.push_back(idx);
      }
      if (!ok || face.size() < 3) { warning += "skipped malformed face\n"; continue; }
      // preconditions the reference leaves unchecked: every corner needs a normal, every face a material
      bool normalsOk = true;
      for (auto& f : face) normalsOk = normalsOk && f.vn >= 0 && f.vn < (int)vn.size() / 3;
      if (!normalsOk) { error = "face without vertex normals (vn) -- required by the renderer"; checkError(); return; }
      if (currentMaterial < 0) { error = "face without usemtl -- required by the renderer"; checkError(); return; }
      std::vector<int> tris;
      if (face.size() == 3) {
        tris = {0, 1, 2};
      } else if (face.size() == 4) {
        // quads split on the shorter diagonal, ties on 1-3 (tinyobjloader's rule, which the reference relies on)
        const float* p0 = &v[3 * face[0].v]; const float* p1 = &v[3 * face[1].v];
        const float* p2 = &v[3 * face[2].v]; const float* p3 = &v[3 * face[3].v];
        float e02[3], e13[3];
        for (int a = 0; a < 3; a++) { e02[a] = p2[a] - p0[a]; e13[a] = p3[a] - p1[a]; }
        const float s02 = e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2];
        const float s13 = e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2];
        if (s02 < s13) tris = {0, 1, 2, 0, 2, 3}; else tris = {0, 1, 3, 1, 2, 3};
      } else {
        std::vector<const float*> pts;
        for (auto& f : face) pts.push_back(&v[3 * f.v]);
        earClip(pts, tris);
      }
      for (size_t t = 0; t + 2 < tris.size(); t += 3) {
        float p[9], n[9];
        for (int c = 0; c < 3; c++) {
          memcpy(p + 3 * c, &v[3 * face[tris[t + c]].v], 12);
          memcpy(n + 3 * c, &vn[3 * face[tris[t + c]].vn], 12);
        }
        addTriangle(p, n, currentMaterial);
      }
    }
  }
  success = !primitiveInfoList.empty() && !materialList.empty();
  if (!success && error.empty()) error = "no triangles or no materials in " + fileName;
  checkError();
}

Model::~Model() {}
std::string Model::getFileName() { return fileName; }
bool Model::checkError() {
  if (!warning.empty()) printf("%s\n", warning.c_str());
  if (!error.empty()) printf("%s\n", error.c_str());
  return success;
}
std::vector<PrimitiveInfo>* Model::getPrimitiveInfoListP() { return &primitiveInfoList; }
uint64_t Model::getMaterialBufferSize() { return sizeof(Material) * materialList.size(); }
void* Model::getMaterialBuffer() { return materialList.data(); }

// ------------------------------------------------------------------------------------------ BVH
namespace {

struct BuildRange { int start, end, node, depth; };

}  // namespace

// Iterative build straight into the flattened pre-order array: node i's left subtree starts at i+1 and holds exactly
// 2*leftCount-1 nodes (every leaf is one triangle), so the right child's index is known before either subtree is built --
// whatever the split position, which is what lets the same loop serve both split rules:
//   ACCELERATION_STRUCTURE_TYPE_BVH      median split of the largest centroid extent, the reference's rule
//                                        (src/acceleration_structure_explicit.cpp:47-137, without its reads of uninitialised
//                                        centroid bounds);
//   ACCELERATION_STRUCTURE_TYPE_BVH_SAH  binned surface-area heuristic: 32 bins per axis over the centroid bounds, the split
//                                        plane (over all three axes) that minimises area(L)*count(L) + area(R)*count(R).
//                                        The traversal stack of the renderers is as deep as the tree is high (the reference's
//                                        is a fixed 64, acc.cl:137; this backend sizes an LDS stack per launch), so the
//                                        height is bounded: a subtree of k triangles at depth d must fit below
//                                        ceil(log2 n) + kSahSlack levels, and a split that would not leave room falls back
//                                        to the median.  Flatten order, axis field (the split axis, which the traversal
//                                        uses for near / far), leaf = one triangle: as above.
namespace {
constexpr int kSahBins = 32;
constexpr int kSahSlack = 4;

int ceilLog2(int x) {
  int l = 0;
  while ((1 << l) < x) l++;
  return l;
}

float halfArea(const float* lo, const float* hi) {
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}
}  // namespace

AccelerationStructureExplicit::AccelerationStructureExplicit(AccelerationStructureExplicitProperties properties) : height(0) {
  static_assert(sizeof(LinearBVHNode) == 32 && sizeof(Primitive) == 76 && sizeof(LightContainer) == 260 && sizeof(Material) == 32,
                "buffer layouts are part of the renderer contract");
  memset(&lightContainer, 0, sizeof(lightContainer));
  Model* pModel = (Model*)properties.pModel;
  std::vector<PrimitiveInfo>& prims = *pModel->getPrimitiveInfoListP();
  const int n = (int)prims.size();
  if (n == 0) return;
  const Material* materials = (const Material*)pModel->getMaterialBuffer();
  const uint64_t materialCount = pModel->getMaterialBufferSize() / sizeof(Material);
  const bool sah = properties.accelerationStructureExplicitType == ACCELERATION_STRUCTURE_TYPE_BVH_SAH;
  const int heightLimit = ceilLog2(n) + kSahSlack;

  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  nodes.resize(2 * (size_t)n - 1);
  memset(nodes.data(), 0, nodes.size() * sizeof(LinearBVHNode));

  std::vector<BuildRange> work;
  work.push_back({0, n, 0, 0});
  while (!work.empty()) {
    const BuildRange r = work.back();
    work.pop_back();
    LinearBVHNode& node = nodes[r.node];
    float cmin[3], cmax[3];
    for (int a = 0; a < 3; a++) {
      node.boundsMin[a] = std::numeric_limits<float>::max();
      node.boundsMax[a] = -std::numeric_limits<float>::max();
      cmin[a] = std::numeric_limits<float>::max();
      cmax[a] = -std::numeric_limits<float>::max();
    }
    for (int i = r.start; i < r.end; i++) {
      const PrimitiveInfo& p = prims[order[i]];
      for (int a = 0; a < 3; a++) {
        node.boundsMin[a] = std::min(node.boundsMin[a], p.boundsMin[a]);
        node.boundsMax[a] = std::max(node.boundsMax[a], p.boundsMax[a]);
        cmin[a] = std::min(cmin[a], p.centroid[a]);
        cmax[a] = std::max(cmax[a], p.centroid[a]);
      }
    }
    height = std::max(height, r.depth);
    const int count = r.end - r.start;
    if (count == 1) {
      node.primitivesOffset = r.start;   // == position in the ordered primitive buffer
      node.primitiveCount = 1;
      node.axis = 0;
      continue;
    }
    // largest centroid extent; ties resolved x, then y, then z (the reference picks x only when strictly largest,
    // then y over z when strictly larger)
    const float d[3] = {cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2]};
    int dim = (d[0] > d[1] && d[0] > d[2]) ? 0 : (d[1] > d[2] ? 1 : 2);
    int mid = (r.start + r.end) / 2;
    bool split = false;
    // a child of k triangles needs ceil(log2 k) more levels at least: both children must fit under the height limit
    const int maxChild = (heightLimit - r.depth - 1) >= 30 ? count : std::min(count, 1 << std::max(0, heightLimit - r.depth - 1));
    if (sah && count > 2 && maxChild >= (count + 1) / 2) {
      float bestCost = std::numeric_limits<float>::max();
      int bestDim = -1, bestBin = -1;
      for (int a = 0; a < 3; a++) {
        if (!(d[a] > 0.0f)) continue;
        struct Bin { float lo[3], hi[3]; int count; } bins[kSahBins];
        for (Bin& b : bins) {
          for (int k = 0; k < 3; k++) { b.lo[k] = std::numeric_limits<float>::max(); b.hi[k] = -std::numeric_limits<float>::max(); }
          b.count = 0;
        }
        const float scale = (float)kSahBins / d[a];
        for (int i = r.start; i < r.end; i++) {
          const PrimitiveInfo& p = prims[order[i]];
          const int b = std::min(kSahBins - 1, std::max(0, (int)((p.centroid[a] - cmin[a]) * scale)));
          bins[b].count++;
          for (int k = 0; k < 3; k++) { bins[b].lo[k] = std::min(bins[b].lo[k], p.boundsMin[k]); bins[b].hi[k] = std::max(bins[b].hi[k], p.boundsMax[k]); }
        }
        // sweep: rightArea[b] = area of bins b..end
        float rightArea[kSahBins];
        int rightCount[kSahBins];
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::max(); hi[k] = -std::numeric_limits<float>::max(); }
        int c = 0;
        for (int b = kSahBins - 1; b >= 0; b--) {
          if (bins[b].count) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], bins[b].lo[k]); hi[k] = std::max(hi[k], bins[b].hi[k]); }
          c += bins[b].count;
          rightCount[b] = c;
          rightArea[b] = c ? halfArea(lo, hi) : 0.0f;
        }
        for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::max(); hi[k] = -std::numeric_limits<float>::max(); }
        c = 0;
        for (int b = 0; b + 1 < kSahBins; b++) {   // split after bin b
          if (bins[b].count) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], bins[b].lo[k]); hi[k] = std::max(hi[k], bins[b].hi[k]); }
          c += bins[b].count;
          const int rc = rightCount[b + 1];
          if (c == 0 || rc == 0 || c > maxChild || rc > maxChild) continue;
          const float cost = halfArea(lo, hi) * (float)c + rightArea[b + 1] * (float)rc;
          if (cost < bestCost) { bestCost = cost; bestDim = a; bestBin = b; }
        }
      }
      if (bestDim >= 0) {
        const float scale = (float)kSahBins / d[bestDim];
        auto it = std::stable_partition(order.begin() + r.start, order.begin() + r.end, [&](int i) {
          return std::min(kSahBins - 1, std::max(0, (int)((prims[i].centroid[bestDim] - cmin[bestDim]) * scale))) <= bestBin;
        });
        mid = (int)(it - order.begin());
        dim = bestDim;
        split = true;
      }
    }
    // median split: the reference's rule, and the fallback of the SAH build (no plane separates the centroids, or the height
    // limit leaves no room).  Coincident centroids on every axis (e.g. the two halves of a quad): still split, by input
    // order, so that no leaf ever holds two triangles
    if (!split && d[dim] > 0.0f) {
      std::nth_element(order.begin() + r.start, order.begin() + mid, order.begin() + r.end, [&](int a, int b) {
        const float ca = prims[a].centroid[dim], cb = prims[b].centroid[dim];
        return ca < cb || (ca == cb && a < b);   // total order: the build is deterministic
      });
    }
    node.axis = (uint8_t)dim;
    node.primitiveCount = 0;
    const int leftCount = mid - r.start;
    node.secondChildOffset = r.node + 2 * leftCount;   // r.node + 1 + (2*leftCount - 1)
    work.push_back({mid, r.end, node.secondChildOffset, r.depth + 1});
    work.push_back({r.start, mid, r.node + 1, r.depth + 1});
  }

  orderedPrimitives.resize(n);
  for (int i = 0; i < n; i++) {
    const PrimitiveInfo& p = prims[order[i]];
    Primitive& q = orderedPrimitives[i];
    memcpy(q.positionA, p.positionA, 12); memcpy(q.positionB, p.positionB, 12); memcpy(q.positionC, p.positionC, 12);
    memcpy(q.normalA, p.normalA, 12); memcpy(q.normalB, p.normalB, 12); memcpy(q.normalC, p.normalC, 12);
    q.materialIndex = p.materialIndex;
    if (p.materialIndex >= 0 && (uint64_t)p.materialIndex < materialCount) {
      const Material& m = materials[p.materialIndex];
      if ((m.emission[0] > 0 || m.emission[1] > 0 || m.emission[2] > 0) && lightContainer.count < 64) {
        lightContainer.primitives[lightContainer.count] = (uint32_t)i;
        lightContainer.count += 1;
      }
    }
  }
}

AccelerationStructureExplicit::~AccelerationStructureExplicit() {}
uint64_t AccelerationStructureExplicit::getNodeBufferSize() { return sizeof(LinearBVHNode) * nodes.size(); }
void* AccelerationStructureExplicit::getNodeBuffer() { return nodes.data(); }
uint64_t AccelerationStructureExplicit::getOrderedPrimitiveBufferSize() { return sizeof(Primitive) * orderedPrimitives.size(); }
void* AccelerationStructureExplicit::getOrderedPrimitiveBuffer() { return orderedPrimitives.data(); }
uint64_t AccelerationStructureExplicit::getLightContainerBufferSize() { return sizeof(LightContainer); }
void* AccelerationStructureExplicit::getLightContainerBuffer() { return &lightContainer; }
int AccelerationStructureExplicit::getHeight() { return height; }

// ------------------------------------------------------------------------------------------ C wrappers (Python / tools)
extern "C" {

struct lt_host_scene {
  Model* model;
  AccelerationStructureExplicit* as;
};

static lt_host_scene* finishScene(Model* m, int type = ACCELERATION_STRUCTURE_TYPE_BVH) {
  if (!m->checkError()) { delete m; return nullptr; }
  AccelerationStructureExplicitProperties props;
  props.sType = STRUCTURE_TYPE_ACCELERATION_STRUCTURE_PROPERTIES;
  props.pNext = nullptr;
  props.accelerationStructureExplicitType = type == ACCELERATION_STRUCTURE_TYPE_BVH_SAH ? ACCELERATION_STRUCTURE_TYPE_BVH_SAH : ACCELERATION_STRUCTURE_TYPE_BVH;
  props.pModel = m;
  lt_host_scene* s = new lt_host_scene;
  s->model = m;
  s->as = new AccelerationStructureExplicit(props);
  return s;
}

lt_host_scene* lt_host_scene_from_obj(const char* path) { return finishScene(new Model(std::string(path))); }

lt_host_scene* lt_host_scene_from_triangles(const float* positions, const float* normals, const int* materialIndices,
                                            uint64_t triangleCount, const void* materials, uint64_t materialCount) {
  return finishScene(new Model(positions, normals, materialIndices, triangleCount, (const Material*)materials, materialCount));
}

// the same with the acceleration-structure type spelled out (ACCELERATION_STRUCTURE_TYPE_BVH = 0, ..._BVH_SAH = 1)
lt_host_scene* lt_host_scene_from_obj_ex(const char* path, int type) { return finishScene(new Model(std::string(path)), type); }

lt_host_scene* lt_host_scene_from_triangles_ex(const float* positions, const float* normals, const int* materialIndices,
                                               uint64_t triangleCount, const void* materials, uint64_t materialCount, int type) {
  return finishScene(new Model(positions, normals, materialIndices, triangleCount, (const Material*)materials, materialCount), type);
}

// which: 0 nodes, 1 ordered primitives, 2 materials, 3 light container
const void* lt_host_scene_buffer(lt_host_scene* s, int which, uint64_t* bytes) {
  switch (which) {
    case 0: *bytes = s->as->getNodeBufferSize(); return s->as->getNodeBuffer();
    case 1: *bytes = s->as->getOrderedPrimitiveBufferSize(); return s->as->getOrderedPrimitiveBuffer();
    case 2: *bytes = s->model->getMaterialBufferSize(); return s->model->getMaterialBuffer();
    case 3: *bytes = s->as->getLightContainerBufferSize(); return s->as->getLightContainerBuffer();
  }
  *bytes = 0;
  return nullptr;
}

int lt_host_scene_height(lt_host_scene* s) { return s->as->getHeight(); }

void lt_host_scene_free(lt_host_scene* s) {
  if (!s) return;
  delete s->as;
  delete s->model;
  delete s;
}

}  // extern "C"
