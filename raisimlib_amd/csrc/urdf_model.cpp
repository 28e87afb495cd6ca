// urdf_model.cpp — host-side model loader: URDF subset -> flat rsb_model_blob (cold path).
//
// Plays the role of raisim::World::addArticulatedSystem(urdfPath) [RECALL; ArticulatedSystem.hpp /
// World.hpp are absent from /root/reference, SURVEY.md §3.3]: parse the robot description once on the
// host, merge fixed joints, and emit the immutable model blob that rsb_create uploads to the device.
//
// Supported subset: <link>/<inertial>/<collision> with <sphere>, <capsule>, <box> (its 8 corners) and <cylinder>
// (the lowest rim point of each end cap) geometry,
// <joint type="revolute|continuous|prismatic|fixed">, <origin xyz rpy>, <axis>, <limit>,
// <dynamics damping rotor_inertia>.  The root link is the floating base, or - when it is named "world" - a fixed base.  <mesh filename=.. scale=..> collision geometry
// (Wavefront OBJ, binary / ASCII STL, Collada .dae; path relative to the URDF file, "package://" / "file://" prefixes stripped to the
// longest existing suffix) becomes a POINT SET: up to kMeshPoints vertices of the mesh's convex hull (support vertices of
// the body diagonals, axes and face diagonals), each a zero-radius sphere - against a plane this is the contact set of a
// triangle-mesh x plane collider (vertices below the plane) thinned out to the budget.  Meshes that cannot be read
// (other formats, missing files, URDFs given as strings without a directory) are skipped and counted in
// rsb_model::skipped_collisions.
#include "rsb.h"
#include "rsb_internal.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace rsb {

static thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }
const char* last_error() { return g_last_error.c_str(); }

// ---------------------------------------------------------------------------- tiny XML reader
struct XmlNode {
  std::string tag;
  std::string text;   // character data between the children (Collada float arrays; URDF has none)
  std::map<std::string, std::string> attr;
  std::vector<std::unique_ptr<XmlNode>> kids;
  const XmlNode* child(const char* t) const {
    for (auto& k : kids) if (k->tag == t) return k.get();
    return nullptr;
  }
  const char* get(const char* a) const {
    auto it = attr.find(a);
    return it == attr.end() ? nullptr : it->second.c_str();
  }
};

class XmlParser {
 public:
  explicit XmlParser(const std::string& s) : s_(s), i_(0) {}
  std::unique_ptr<XmlNode> parse() {
    skip_misc();
    auto n = element();
    if (!n) throw std::runtime_error("URDF: no root element");
    return n;
  }

 private:
  const std::string& s_;
  size_t i_;
  bool starts(const char* p) const { return s_.compare(i_, std::strlen(p), p) == 0; }
  void skip_ws() { while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_; }
  void skip_until(const char* end) {
    size_t p = s_.find(end, i_);
    if (p == std::string::npos) throw std::runtime_error(std::string("URDF: unterminated '") + end + "'");
    i_ = p + std::strlen(end);
  }
  void skip_misc() {
    for (;;) {
      skip_ws();
      if (starts("<?")) skip_until("?>");
      else if (starts("<!--")) skip_until("-->");
      else if (starts("<!")) skip_until(">");
      else break;
    }
  }
  std::string name() {
    size_t b = i_;
    while (i_ < s_.size() && (std::isalnum((unsigned char)s_[i_]) || s_[i_] == '_' || s_[i_] == ':' || s_[i_] == '-' || s_[i_] == '.')) ++i_;
    if (b == i_) throw std::runtime_error("URDF: expected a name at offset " + std::to_string(b));
    return s_.substr(b, i_ - b);
  }
  std::unique_ptr<XmlNode> element() {
    if (i_ >= s_.size() || s_[i_] != '<') return nullptr;
    ++i_;
    auto n = std::make_unique<XmlNode>();
    n->tag = name();
    for (;;) {
      skip_ws();
      if (i_ >= s_.size()) throw std::runtime_error("URDF: unexpected end in <" + n->tag + ">");
      if (starts("/>")) { i_ += 2; return n; }
      if (s_[i_] == '>') { ++i_; break; }
      std::string a = name();
      skip_ws();
      if (i_ >= s_.size() || s_[i_] != '=') throw std::runtime_error("URDF: expected '=' after attribute " + a);
      ++i_;
      skip_ws();
      char qc = s_[i_];
      if (qc != '"' && qc != '\'') throw std::runtime_error("URDF: unquoted attribute " + a);
      size_t e = s_.find(qc, i_ + 1);
      if (e == std::string::npos) throw std::runtime_error("URDF: unterminated attribute " + a);
      n->attr[a] = s_.substr(i_ + 1, e - i_ - 1);
      i_ = e + 1;
    }
    for (;;) {  // children / text until </tag>
      size_t lt = s_.find('<', i_);
      if (lt == std::string::npos) throw std::runtime_error("URDF: missing </" + n->tag + ">");
      if (lt > i_) n->text.append(s_, i_, lt - i_);
      i_ = lt;
      if (starts("<!--")) { skip_until("-->"); continue; }
      if (starts("<?")) { skip_until("?>"); continue; }
      if (starts("</")) {
        i_ += 2;
        std::string t = name();
        if (t != n->tag) throw std::runtime_error("URDF: </" + t + "> closes <" + n->tag + ">");
        skip_ws();
        if (i_ >= s_.size() || s_[i_] != '>') throw std::runtime_error("URDF: malformed </" + t + ">");
        ++i_;
        return n;
      }
      n->kids.push_back(element());
    }
  }
};

// ------------------------------------------------------------------------------- small math
struct V3 { double x = 0, y = 0, z = 0; };
struct M3 { double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; };
struct Xf { M3 R; V3 p; };  // parent <- child

static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static V3 mul(const M3& A, V3 v) {
  return {A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
          A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
static M3 mul(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
static M3 transpose(const M3& A) {
  M3 T;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T.m[3 * i + j] = A.m[3 * j + i];
  return T;
}
static Xf compose(const Xf& a, const Xf& b) { return {mul(a.R, b.R), a.p + mul(a.R, b.p)}; }
// URDF rpy = fixed-axis roll(x), pitch(y), yaw(z): R = Rz(y) Ry(p) Rx(r)
static M3 rpy(double r, double p, double y) {
  double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p), cy = std::cos(y), sy = std::sin(y);
  M3 R;
  R.m[0] = cy * cp; R.m[1] = cy * sp * sr - sy * cr; R.m[2] = cy * sp * cr + sy * sr;
  R.m[3] = sy * cp; R.m[4] = sy * sp * sr + cy * cr; R.m[5] = sy * sp * cr - cy * sr;
  R.m[6] = -sp;     R.m[7] = cp * sr;                R.m[8] = cp * cr;
  return R;
}
static bool parse_doubles(const char* s, double* out, int n) {
  if (!s) return false;
  char* e;
  for (int i = 0; i < n; ++i) {
    out[i] = std::strtod(s, &e);
    if (e == s) return false;
    s = e;
  }
  return true;
}
static double attr_double(const XmlNode* n, const char* a, double dflt) {
  double v;
  if (n && parse_doubles(n->get(a), &v, 1)) return v;
  return dflt;
}
static Xf parse_origin(const XmlNode* n) {
  Xf x;
  if (!n) return x;
  double v[3];
  if (parse_doubles(n->get("xyz"), v, 3)) x.p = {v[0], v[1], v[2]};
  if (parse_doubles(n->get("rpy"), v, 3)) x.R = rpy(v[0], v[1], v[2]);
  return x;
}

// ------------------------------------------------------------------------------ URDF -> blob
struct UCollision { Xf x; int type; double radius, length; double size[3]; std::string name, material; std::vector<V3> pts; };  // type 0 sphere, 1 capsule, 2 box, 3 mesh point set, 4 cylinder (two rim primitives)
struct ULink {
  std::string name;
  double mass = 0;
  Xf inertial;         // link <- inertial frame
  double I[6] = {0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz in the inertial frame
  std::vector<UCollision> cols;
  std::vector<int> child_joints;
  int parent_joint = -1;
};
struct UJoint {
  std::string name, parent, child;
  int type = 0;  // 0 fixed, 1 revolute/continuous, 2 prismatic
  Xf origin;
  V3 axis{1, 0, 0};
  double lower = -1e30, upper = 1e30, effort = 0, damping = 0, armature = 0;
};


// ------------------------------------------------------------------------------ mesh colliders
constexpr int kMeshPoints = 8;   // collision points kept per mesh by default (the model holds RSB_MAX_COLLISIONS primitives in total)
constexpr int kMeshPointsMax = 26;   // ... and at most: one support vertex per direction of thin_point_set (8 body diagonals, 6 axes, 12 face diagonals)
static int g_mesh_points = kMeshPoints;   // rsb_set_mesh_point_budget (process-wide; read when a URDF is loaded)

static bool file_exists(const std::string& p) { std::ifstream f(p, std::ios::binary); return (bool)f; }

// URDF mesh URI -> a readable path: as given (absolute or relative to the URDF's directory), else the longest suffix of a
// package:// / file:// URI that exists below the URDF's directory or one of its parents (ROS package layouts)
static std::string resolve_mesh_path(const std::string& uri, const std::string& base_dir) {
  std::string rel = uri;
  for (const char* pre : {"package://", "file://", "model://"}) if (rel.rfind(pre, 0) == 0) rel = rel.substr(std::strlen(pre));
  // a URI must not walk out of the directories probed below ("../../etc/.."): the probing itself climbs at most 4 parents of the URDF
  for (size_t at = 0; at <= rel.size();) {
    const size_t end = std::min(rel.find('/', at), rel.size());
    if (rel.compare(at, end - at, "..") == 0) return "";
    at = end + 1;
  }
  if (!rel.empty() && rel[0] == '/' && file_exists(rel)) return rel;
  std::string dir = base_dir;
  for (int up = 0; up < 4; ++up) {
    std::string sub = rel;
    while (true) {
      const std::string cand = (dir.empty() ? std::string(".") : dir) + "/" + sub;
      if (file_exists(cand)) return cand;
      const size_t cut = sub.find('/');
      if (cut == std::string::npos) break;
      sub = sub.substr(cut + 1);
    }
    dir += "/..";
  }
  return "";
}

// vertices of an OBJ ("v x y z" lines), an STL (binary: 80-byte header, uint32 count, 50-byte facets; ASCII: "vertex x y z")
// or a Collada file (.dae: the POSITION float arrays of its meshes)
static bool read_mesh_vertices(const std::string& path, std::vector<V3>* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const size_t dot = path.rfind('.');
  std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
  for (auto& ch : ext) ch = (char)std::tolower((unsigned char)ch);
  out->clear();
  if (ext == "obj") {
    std::istringstream ss(data);
    std::string line;
    while (std::getline(ss, line)) {
      if (line.size() > 2 && line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) {
        double v[3];
        if (std::sscanf(line.c_str() + 2, "%lf %lf %lf", &v[0], &v[1], &v[2]) == 3) out->push_back({v[0], v[1], v[2]});
      }
    }
  } else if (ext == "stl") {
    uint32_t nt = 0;
    if (data.size() >= 84) std::memcpy(&nt, data.data() + 80, 4);
    if (data.size() >= 84 && data.size() == 84 + (size_t)nt * 50) {        // binary
      for (uint32_t t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) {
          float v[3];
          std::memcpy(v, data.data() + 84 + (size_t)t * 50 + 12 + 12 * k, 12);
          out->push_back({v[0], v[1], v[2]});
        }
    } else {                                                               // ASCII
      size_t pos = 0;
      while ((pos = data.find("vertex", pos)) != std::string::npos) {
        double v[3];
        if (std::sscanf(data.c_str() + pos + 6, "%lf %lf %lf", &v[0], &v[1], &v[2]) == 3) out->push_back({v[0], v[1], v[2]});
        pos += 6;
      }
    }
  } else if (ext == "dae") {
    // Collada: the POSITION source of every <mesh> - <vertices><input semantic="POSITION" source="#id"/> names the <source> whose
    // <float_array> holds x y z triples - placed by the node transforms of the visual scene (<matrix>, <translate>, <rotate>, <scale> of the
    // <node> chain down to each <instance_geometry url="#geometry">; a geometry no node instantiates is taken as it is).  <unit meter="..">
    // scales to metres; <up_axis>Y_UP</up_axis> is turned to the URDF's z-up.
    std::unique_ptr<XmlNode> root;
    try { XmlParser parser(data); root = parser.parse(); } catch (const std::exception&) { return false; }
    if (!root) return false;
    struct M4 { double m[16]; };
    auto ident = [] { M4 r; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return r; };
    auto mul4 = [](const M4& A, const M4& B) { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double t = 0; for (int k = 0; k < 4; ++k) t += A.m[4 * i + k] * B.m[4 * k + j]; r.m[4 * i + j] = t; } return r; };
    auto numbers = [](const std::string& t) { std::vector<double> f; const char* c = t.c_str(); for (;;) { char* nx = nullptr; const double v = std::strtod(c, &nx); if (nx == c) break; f.push_back(v); c = nx; } return f; };
    double unit = 1.0;
    bool y_up = false;
    if (const XmlNode* as = root->child("asset")) {
      if (const XmlNode* un = as->child("unit")) { if (const char* mt = un->get("meter")) unit = std::atof(mt); }
      if (const XmlNode* up = as->child("up_axis")) y_up = up->text.find("Y_UP") != std::string::npos;
    }
    // geometry id -> its meshes' POSITION triples
    std::map<std::string, std::vector<V3>> geo;
    std::vector<std::string> geo_order;
    for (auto& lib : root->kids) {
      if (lib->tag != "library_geometries") continue;
      for (auto& g : lib->kids) {
        if (g->tag != "geometry") continue;
        const char* gid = g->get("id");
        std::vector<V3> pts;
        for (auto& mesh : g->kids) {
          if (mesh->tag != "mesh") continue;
          for (auto& vs : mesh->kids) {
            if (vs->tag != "vertices") continue;
            std::string src;
            for (auto& in : vs->kids) if (in->tag == "input" && in->get("semantic") && std::string(in->get("semantic")) == "POSITION" && in->get("source")) src = in->get("source");
            if (src.size() < 2 || src[0] != '#') continue;
            for (auto& so : mesh->kids) {
              if (so->tag != "source" || !so->get("id") || src.substr(1) != so->get("id")) continue;
              if (const XmlNode* fa = so->child("float_array")) {
                const std::vector<double> f = numbers(fa->text);
                for (size_t i = 0; i + 2 < f.size(); i += 3) pts.push_back({f[i], f[i + 1], f[i + 2]});
              }
            }
          }
        }
        const std::string key = gid ? gid : ("#" + std::to_string(geo.size()));
        geo[key] = pts; geo_order.push_back(key);
      }
    }
    // the scene's nodes: accumulated transform down to every <instance_geometry>
    std::vector<std::pair<std::string, M4>> inst;
    std::function<void(const XmlNode*, const M4&)> walk = [&](const XmlNode* nd, const M4& parent) {
      M4 T = parent;
      for (auto& k : nd->kids) {
        if (k->tag == "matrix") { const auto f = numbers(k->text); if (f.size() >= 16) { M4 L; for (int i = 0; i < 16; ++i) L.m[i] = f[i]; T = mul4(T, L); } }
        else if (k->tag == "translate") { const auto f = numbers(k->text); if (f.size() >= 3) { M4 L = ident(); L.m[3] = f[0]; L.m[7] = f[1]; L.m[11] = f[2]; T = mul4(T, L); } }
        else if (k->tag == "scale") { const auto f = numbers(k->text); if (f.size() >= 3) { M4 L = ident(); L.m[0] = f[0]; L.m[5] = f[1]; L.m[10] = f[2]; T = mul4(T, L); } }
        else if (k->tag == "rotate") {
          const auto f = numbers(k->text);
          if (f.size() >= 4) {
            const double n = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
            if (n > 1e-12) {
              const double x = f[0] / n, y = f[1] / n, z = f[2] / n, an = f[3] * 3.14159265358979323846 / 180.0, c = std::cos(an), sn = std::sin(an), t = 1 - c;
              M4 L = ident();
              L.m[0] = t * x * x + c;      L.m[1] = t * x * y - sn * z; L.m[2] = t * x * z + sn * y;
              L.m[4] = t * x * y + sn * z; L.m[5] = t * y * y + c;      L.m[6] = t * y * z - sn * x;
              L.m[8] = t * x * z - sn * y; L.m[9] = t * y * z + sn * x; L.m[10] = t * z * z + c;
              T = mul4(T, L);
            }
          }
        }
      }
      for (auto& k : nd->kids) {
        if (k->tag == "instance_geometry" && k->get("url") && k->get("url")[0] == '#') inst.push_back({std::string(k->get("url") + 1), T});
        else if (k->tag == "node") walk(k.get(), T);
      }
    };
    for (auto& lib : root->kids)
      if (lib->tag == "library_visual_scenes")
        for (auto& sc : lib->kids)
          if (sc->tag == "visual_scene")
            for (auto& nd : sc->kids)
              if (nd->tag == "node") walk(nd.get(), ident());
    auto emit = [&](const std::vector<V3>& pts, const M4& T) {
      for (const V3& v : pts) {
        const double x = (T.m[0] * v.x + T.m[1] * v.y + T.m[2] * v.z + T.m[3]) * unit, y = (T.m[4] * v.x + T.m[5] * v.y + T.m[6] * v.z + T.m[7]) * unit,
                     z = (T.m[8] * v.x + T.m[9] * v.y + T.m[10] * v.z + T.m[11]) * unit;
        if (y_up) out->push_back({x, -z, y}); else out->push_back({x, y, z});
      }
    };
    std::map<std::string, int> used;
    for (auto& in : inst) { auto it = geo.find(in.first); if (it != geo.end()) { emit(it->second, in.second); ++used[in.first]; } }
    for (auto& key : geo_order) if (!used.count(key)) emit(geo[key], ident());
  } else {
    return false;
  }
  return !out->empty();
}

// at most `budget` vertices of the cloud's CONVEX HULL: the support vertices (farthest along a direction) of the 8 body
// diagonals, then the 6 axes, then the 12 face diagonals - a box keeps exactly its corners, a foot plate its outline
static std::vector<V3> thin_point_set(const std::vector<V3>& v, int budget) {
  std::vector<V3> sel;
  std::vector<V3> dirs;
  for (int sx = -1; sx <= 1; sx += 2) for (int sy = -1; sy <= 1; sy += 2) for (int sz = -1; sz <= 1; sz += 2) dirs.push_back({(double)sx, (double)sy, (double)sz});
  for (int ax = 0; ax < 3; ++ax) for (int sg = -1; sg <= 1; sg += 2) dirs.push_back({ax == 0 ? (double)sg : 0.0, ax == 1 ? (double)sg : 0.0, ax == 2 ? (double)sg : 0.0});
  for (int a = 0; a < 3; ++a) for (int sa = -1; sa <= 1; sa += 2) for (int sb = -1; sb <= 1; sb += 2) {
    double d[3] = {0, 0, 0};
    d[a] = sa; d[(a + 1) % 3] = sb;
    dirs.push_back({d[0], d[1], d[2]});
  }
  for (const V3& d : dirs) {
    if ((int)sel.size() >= budget) break;
    size_t best = 0;
    double sb = -1e300;
    for (size_t i = 0; i < v.size(); ++i) {
      const double sp = d.x * v[i].x + d.y * v[i].y + d.z * v[i].z;
      if (sp > sb) { sb = sp; best = i; }
    }
    bool dup = false;
    for (auto& q : sel) { V3 e = q - v[best]; if (e.x * e.x + e.y * e.y + e.z * e.z < 1e-18) dup = true; }
    if (!dup) sel.push_back(v[best]);
  }
  return sel;
}

struct Builder {
  std::vector<ULink> links;
  std::vector<UJoint> joints;
  std::map<std::string, int> link_ix;
  rsb_model_blob blob;
  int skipped_collisions = 0;
  double sample_spacing = 0.0;   // > 0: capsules and boxes also get sample primitives on their surface (rsb_model_from_urdf_*_sampled)

  // accumulated rigid body (a moving link + everything fixed to it)
  struct Acc { double m = 0; V3 mc; std::vector<std::pair<double, std::pair<V3, M3>>> parts; };
  std::vector<Acc> acc;

  void add_inertial(int body, const Xf& body_from_link, const ULink& L) {
    if (L.mass <= 0) return;
    Xf bi = compose(body_from_link, L.inertial);
    M3 Il;
    Il.m[0] = L.I[0]; Il.m[1] = L.I[1]; Il.m[2] = L.I[2];
    Il.m[3] = L.I[1]; Il.m[4] = L.I[3]; Il.m[5] = L.I[4];
    Il.m[6] = L.I[2]; Il.m[7] = L.I[4]; Il.m[8] = L.I[5];
    M3 Ib = mul(mul(bi.R, Il), transpose(bi.R));
    acc[body].m += L.mass;
    acc[body].mc = acc[body].mc + L.mass * bi.p;
    acc[body].parts.push_back({L.mass, {bi.p, Ib}});
  }
  void add_collisions(int body, const Xf& body_from_link, const ULink& L) {
    for (auto& c : L.cols) {
      Xf bc = compose(body_from_link, c.x);
      int n = (c.type == 1 || c.type == 4) ? 2 : (c.type == 2 ? 8 : (c.type == 3 ? (int)c.pts.size() : 1));
      // Sampled colliders (opt-in): the end spheres of a capsule and the corners of a box are its exact contact set on a PLANE only.
      // With a sample spacing h, a capsule also gets spheres of its radius along its axis and a box zero-radius points on the
      // lattice of its edges and faces, both no further apart than h: against a height field (a ridge under the middle of a
      // capsule, a bump under a box face) and against the other links (capsule x capsule) the contact is then found to within h -
      // by the same sphere tests, no new narrow phase in the kernel.  More primitives, more (redundant) contacts on a plane.
      std::vector<V3> extra;
      if (sample_spacing > 0.0 && c.type == 1 && c.length > sample_spacing) {
        const int seg = (int)std::ceil(c.length / sample_spacing);
        for (int q = 1; q < seg; ++q) extra.push_back({0, 0, (0.5 - (double)q / seg) * c.length});
      }
      if (sample_spacing > 0.0 && c.type == 2) {
        int nn[3];
        for (int a = 0; a < 3; ++a) nn[a] = std::max(1, (int)std::ceil(c.size[a] / sample_spacing));
        for (int i = 0; i <= nn[0]; ++i) for (int j = 0; j <= nn[1]; ++j) for (int k2 = 0; k2 <= nn[2]; ++k2) {
          const int ext = (i == 0 || i == nn[0]) + (j == 0 || j == nn[1]) + (k2 == 0 || k2 == nn[2]);
          if (ext == 0 || ext == 3) continue;      // inside the box / a corner (already a primitive)
          extra.push_back({((double)i / nn[0] - 0.5) * c.size[0], ((double)j / nn[1] - 0.5) * c.size[1], ((double)k2 / nn[2] - 0.5) * c.size[2]});
        }
      }
      for (int e = 0; e < n + (int)extra.size(); ++e) {
        if (blob.ncol >= RSB_MAX_COLLISIONS) throw std::runtime_error("URDF: more than RSB_MAX_COLLISIONS collision primitives" + std::string(sample_spacing > 0.0 ? " (sampled colliders: use a larger spacing)" : ""));
        if (e >= n) {   // a sample primitive
          V3 p = bc.p + mul(bc.R, extra[e - n]);
          int s2 = blob.ncol++;
          blob.col_body[s2] = body;
          blob.col_pos[s2][0] = p.x; blob.col_pos[s2][1] = p.y; blob.col_pos[s2][2] = p.z;
          blob.col_radius[s2] = c.type == 1 ? c.radius : 0.0;
          std::string nm2 = (c.name.empty() ? L.name : c.name) + "/s" + std::to_string(e - n);
          std::snprintf(blob.col_name[s2], RSB_NAME_LEN, "%s", nm2.c_str());
          std::snprintf(blob.col_material[s2], RSB_NAME_LEN, "%s", c.material.empty() ? "default" : c.material.c_str());
          continue;
        }
        V3 off{0, 0, 0};
        if (c.type == 1 || c.type == 4) off = {0, 0, (e == 0 ? 0.5 : -0.5) * c.length};
        if (c.type == 2) off = {(e & 1 ? 0.5 : -0.5) * c.size[0], (e & 2 ? 0.5 : -0.5) * c.size[1], (e & 4 ? 0.5 : -0.5) * c.size[2]};
        if (c.type == 3) off = c.pts[e];
        V3 p = bc.p + mul(bc.R, off);
        int s = blob.ncol++;
        blob.col_body[s] = body;
        blob.col_pos[s][0] = p.x; blob.col_pos[s][1] = p.y; blob.col_pos[s][2] = p.z;
        blob.col_radius[s] = c.type == 4 ? 0.0 : c.radius;
        if ((c.type == 1 || c.type == 4) && e == 0) blob.col_capsule[s] = s + 2;      // the capsule's / cylinder's other end is emitted next (rsb_model_blob::col_capsule)
        if (c.type == 2 && e == 0) blob.col_capsule[s] = -1;                           // the first of a box's eight corners (emitted in bit order)
        if (c.type == 4) {   // end cap of a cylinder: the lowest point of its rim (see rsb_model_blob::col_rim)
          V3 ax = mul(bc.R, V3{0, 0, 1});
          blob.col_axis[s][0] = ax.x; blob.col_axis[s][1] = ax.y; blob.col_axis[s][2] = ax.z;
          blob.col_rim[s] = c.radius;
        }
        std::string nm = c.name.empty() ? L.name : c.name;
        if (c.type == 1 || c.type == 4) nm += (e == 0 ? "/top" : "/bottom");
        if (c.type == 2) nm += "/c" + std::to_string(e);
        if (c.type == 3) nm += "/m" + std::to_string(e);
        std::snprintf(blob.col_name[s], RSB_NAME_LEN, "%s", nm.c_str());
        std::snprintf(blob.col_material[s], RSB_NAME_LEN, "%s", c.material.empty() ? "default" : c.material.c_str());
      }
    }
  }
  // depth-first over the URDF tree; `body` is the moving body `link` belongs to
  void visit(int link, int body, const Xf& body_from_link) {
    const ULink& L = links[link];
    add_inertial(body, body_from_link, L);
    add_collisions(body, body_from_link, L);
    for (int jx : L.child_joints) {
      const UJoint& J = joints[jx];
      int child = link_ix.at(J.child);
      Xf body_from_joint = compose(body_from_link, J.origin);
      if (J.type == 0) { visit(child, body, body_from_joint); continue; }
      if (blob.nb >= RSB_MAX_BODIES) throw std::runtime_error("URDF: more than RSB_MAX_BODIES moving bodies");
      int b = blob.nb++;
      acc.emplace_back();
      blob.parent[b] = body;
      blob.level[b] = blob.level[body] + 1;
      if (blob.level[b] + 1 > blob.depth) blob.depth = blob.level[b] + 1;
      blob.jtype[b] = J.type == 1 ? RSB_JOINT_REVOLUTE : RSB_JOINT_PRISMATIC;
      double an = std::sqrt(J.axis.x * J.axis.x + J.axis.y * J.axis.y + J.axis.z * J.axis.z);
      if (an < 1e-12) throw std::runtime_error("URDF: zero joint axis on " + J.name);
      blob.axis[b][0] = J.axis.x / an; blob.axis[b][1] = J.axis.y / an; blob.axis[b][2] = J.axis.z / an;
      blob.ptree[b][0] = body_from_joint.p.x; blob.ptree[b][1] = body_from_joint.p.y; blob.ptree[b][2] = body_from_joint.p.z;
      for (int c = 0; c < 9; ++c) blob.rtree[b][c] = body_from_joint.R.m[c];
      blob.armature[b] = J.armature; blob.damping[b] = J.damping;
      blob.q_lower[b] = J.lower; blob.q_upper[b] = J.upper; blob.effort[b] = J.effort;
      std::snprintf(blob.body_name[b], RSB_NAME_LEN, "%s", links[child].name.c_str());
      std::snprintf(blob.joint_name[b], RSB_NAME_LEN, "%s", J.name.c_str());
      visit(child, b, Xf{});
    }
  }
  void finish_inertia() {
    for (int b = 0; b < blob.nb; ++b) {
      Acc& a = acc[b];
      if (a.m <= 0 && b == 0 && blob.fixed_base) {   // a massless "world" root: any positive inertia will do, it is never used as a finite one
        blob.mass[0] = 1.0; blob.inertia[0][0] = blob.inertia[0][3] = blob.inertia[0][5] = 1.0;
        continue;
      }
      if (a.m <= 0) throw std::runtime_error(std::string("URDF: moving body '") + blob.body_name[b] + "' has no mass");
      V3 c = (1.0 / a.m) * a.mc;
      double I[9] = {0};
      for (auto& part : a.parts) {
        double mk = part.first;
        V3 d = part.second.first - c;
        const M3& Ik = part.second.second;
        double dd = d.x * d.x + d.y * d.y + d.z * d.z, dv[3] = {d.x, d.y, d.z};
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) I[3 * i + j] += Ik.m[3 * i + j] + mk * ((i == j ? dd : 0.0) - dv[i] * dv[j]);
      }
      blob.mass[b] = a.m;
      blob.com[b][0] = c.x; blob.com[b][1] = c.y; blob.com[b][2] = c.z;
      blob.inertia[b][0] = I[0]; blob.inertia[b][1] = I[1]; blob.inertia[b][2] = I[2];
      blob.inertia[b][3] = I[4]; blob.inertia[b][4] = I[5]; blob.inertia[b][5] = I[8];
    }
  }
};

static void build_from_xml(const std::string& xml, rsb_model_blob* out, int* skipped, const std::string& base_dir = std::string(), double sample_spacing = 0.0) {
  XmlParser parser(xml);
  auto root = parser.parse();
  if (root->tag != "robot") throw std::runtime_error("URDF: root element is <" + root->tag + ">, expected <robot>");
  Builder B;
  B.sample_spacing = sample_spacing;
  std::memset(&B.blob, 0, sizeof B.blob);
  for (auto& k : root->kids) {
    if (k->tag == "link") {
      ULink L;
      const char* nm = k->get("name");
      if (!nm) throw std::runtime_error("URDF: <link> without name");
      L.name = nm;
      if (const XmlNode* in = k->child("inertial")) {
        L.inertial = parse_origin(in->child("origin"));
        L.mass = attr_double(in->child("mass"), "value", 0.0);
        const XmlNode* I = in->child("inertia");
        static const char* keys[6] = {"ixx", "ixy", "ixz", "iyy", "iyz", "izz"};
        for (int c = 0; c < 6; ++c) L.I[c] = attr_double(I, keys[c], 0.0);
      }
      for (auto& c : k->kids) {
        if (c->tag != "collision") continue;
        const XmlNode* g = c->child("geometry");
        if (!g) continue;
        UCollision col;
        col.x = parse_origin(c->child("origin"));
        if (const char* cn = c->get("name")) col.name = cn;
        if (const XmlNode* mt = c->child("material")) { if (const char* mn = mt->get("name")) col.material = mn; }   // RaiSim reads the contact material from here [RECALL]
        if (const XmlNode* s = g->child("sphere")) {
          col.type = 0; col.radius = attr_double(s, "radius", 0.0); col.length = 0;
        } else if (const XmlNode* cp = g->child("capsule")) {
          col.type = 1; col.radius = attr_double(cp, "radius", 0.0); col.length = attr_double(cp, "length", 0.0);
        } else if (const XmlNode* bx = g->child("box")) {
          // a box touches a plane / height field with its corners: 8 zero-radius spheres (exact contact set on a plane)
          col.type = 2; col.radius = 0; col.length = 0;
          if (!parse_doubles(bx->get("size"), col.size, 3) || !(col.size[0] > 0 && col.size[1] > 0 && col.size[2] > 0))
            throw std::runtime_error("URDF: <box> needs size=\"x y z\" > 0 on link " + L.name);
        } else if (const XmlNode* cy = g->child("cylinder")) {
          // a cylinder touches a plane with the lowest points of its two end-cap rims: two rim primitives (exact on a plane,
          // lying, standing or tilted; on a height field the rim point is chosen by the terrain normal under the cap's centre)
          col.type = 4; col.radius = attr_double(cy, "radius", 0.0);
          col.length = attr_double(cy, "length", 0.0);
          if (!(col.length > 0)) throw std::runtime_error("URDF: <cylinder> needs length > 0 on link " + L.name);
        } else if (const XmlNode* ms = g->child("mesh")) {
          // a mesh touches a plane with its vertices: a thinned-out point set of zero-radius spheres (see the file header)
          std::vector<V3> verts;
          const char* fn = ms->get("filename");
          const std::string path = fn ? resolve_mesh_path(fn, base_dir) : std::string();
          if (path.empty() || !read_mesh_vertices(path, &verts)) { ++B.skipped_collisions; continue; }
          double sc[3] = {1, 1, 1};
          if (ms->get("scale") && !parse_doubles(ms->get("scale"), sc, 3)) throw std::runtime_error("URDF: bad <mesh scale> on link " + L.name);
          for (auto& v : verts) v = {v.x * sc[0], v.y * sc[1], v.z * sc[2]};
          col.type = 3; col.radius = 0; col.length = 0;
          col.pts = thin_point_set(verts, g_mesh_points);
        } else { ++B.skipped_collisions; continue; }   // unknown geometry
        if (col.type != 2 && col.type != 3 && col.radius <= 0) throw std::runtime_error("URDF: non-positive collision radius on link " + L.name);
        L.cols.push_back(col);
      }
      if (B.link_ix.count(L.name)) throw std::runtime_error("URDF: duplicate link " + L.name);
      B.link_ix[L.name] = (int)B.links.size();
      B.links.push_back(std::move(L));
    } else if (k->tag == "joint") {
      UJoint J;
      const char* nm = k->get("name");
      const char* ty = k->get("type");
      if (!nm || !ty) throw std::runtime_error("URDF: <joint> needs name and type");
      J.name = nm;
      std::string t = ty;
      if (t == "fixed") J.type = 0;
      else if (t == "revolute" || t == "continuous") J.type = 1;
      else if (t == "prismatic") J.type = 2;
      else throw std::runtime_error("URDF: unsupported joint type '" + t + "' on " + J.name);
      const XmlNode* p = k->child("parent");
      const XmlNode* c = k->child("child");
      if (!p || !c || !p->get("link") || !c->get("link")) throw std::runtime_error("URDF: joint " + J.name + " needs <parent link> and <child link>");
      J.parent = p->get("link"); J.child = c->get("link");
      J.origin = parse_origin(k->child("origin"));
      double v[3];
      if (const XmlNode* ax = k->child("axis")) if (parse_doubles(ax->get("xyz"), v, 3)) J.axis = {v[0], v[1], v[2]};
      if (const XmlNode* lim = k->child("limit")) {
        if (t != "continuous") { J.lower = attr_double(lim, "lower", -1e30); J.upper = attr_double(lim, "upper", 1e30); }
        J.effort = attr_double(lim, "effort", 0.0);
      }
      if (const XmlNode* dyn = k->child("dynamics")) {
        J.damping = attr_double(dyn, "damping", 0.0);
        J.armature = attr_double(dyn, "rotor_inertia", 0.0);
      }
      B.joints.push_back(std::move(J));
    }
  }
  if (B.links.empty()) throw std::runtime_error("URDF: no links");
  for (size_t j = 0; j < B.joints.size(); ++j) {
    auto& J = B.joints[j];
    if (!B.link_ix.count(J.parent) || !B.link_ix.count(J.child)) throw std::runtime_error("URDF: joint " + J.name + " references an unknown link");
    ULink& ch = B.links[B.link_ix[J.child]];
    if (ch.parent_joint >= 0) throw std::runtime_error("URDF: link " + ch.name + " has two parent joints (closed loops unsupported)");
    ch.parent_joint = (int)j;
    B.links[B.link_ix[J.parent]].child_joints.push_back((int)j);
  }
  int rootl = -1;
  for (size_t l = 0; l < B.links.size(); ++l)
    if (B.links[l].parent_joint < 0) {
      if (rootl >= 0) throw std::runtime_error("URDF: more than one root link (" + B.links[rootl].name + ", " + B.links[l].name + ")");
      rootl = (int)l;
    }
  if (rootl < 0) throw std::runtime_error("URDF: no root link");
  // RaiSim's convention [RECALL]: a root link named "world" makes the system fixed-base.  The root body then never moves (the
  // step treats its inertia as infinite); it keeps the 7 + 6 base entries of gc / gv, which stay at the identity pose and zero.
  B.blob.fixed_base = B.links[rootl].name == "world" ? 1 : 0;
  B.blob.nb = 1; B.blob.depth = 1;
  B.blob.parent[0] = -1; B.blob.level[0] = 0; B.blob.jtype[0] = RSB_JOINT_FLOATING;
  B.blob.q_lower[0] = -1e30; B.blob.q_upper[0] = 1e30;
  std::snprintf(B.blob.body_name[0], RSB_NAME_LEN, "%s", B.links[rootl].name.c_str());
  std::snprintf(B.blob.joint_name[0], RSB_NAME_LEN, "base");
  B.acc.emplace_back();
  B.visit(rootl, 0, Xf{});
  B.finish_inertia();
  B.blob.nq = 7 + B.blob.nb - 1;
  B.blob.nv = 6 + B.blob.nb - 1;
  *out = B.blob;
  *skipped = B.skipped_collisions;
}

int validate_blob(const rsb_model_blob& b) {
  if (b.nb < 1 || b.nb > RSB_MAX_BODIES) { set_error("model: nb out of range"); return RSB_E_INVALID; }
  if (b.nq != 7 + b.nb - 1 || b.nv != 6 + b.nb - 1) { set_error("model: nq/nv inconsistent with nb"); return RSB_E_INVALID; }
  if (b.ncol < 0 || b.ncol > RSB_MAX_COLLISIONS) { set_error("model: ncol out of range"); return RSB_E_INVALID; }
  if (b.parent[0] != -1 || b.jtype[0] != RSB_JOINT_FLOATING) { set_error("model: body 0 must be the floating base"); return RSB_E_INVALID; }
  int depth = 1;
  for (int i = 1; i < b.nb; ++i) {
    if (b.parent[i] < 0 || b.parent[i] >= i) { set_error("model: parent[i] must be in [0, i)"); return RSB_E_INVALID; }
    if (b.level[i] != b.level[b.parent[i]] + 1) { set_error("model: level[] inconsistent with parent[]"); return RSB_E_INVALID; }
    if (b.jtype[i] != RSB_JOINT_REVOLUTE && b.jtype[i] != RSB_JOINT_PRISMATIC) { set_error("model: unsupported joint type"); return RSB_E_UNSUPPORTED; }
    if (!(b.mass[i] > 0)) { set_error("model: non-positive body mass"); return RSB_E_INVALID; }
    if (b.level[i] + 1 > depth) depth = b.level[i] + 1;
  }
  if (b.depth != depth) { set_error("model: depth inconsistent"); return RSB_E_INVALID; }
  if (b.fixed_base != 0 && b.fixed_base != 1) { set_error("model: fixed_base must be 0 or 1"); return RSB_E_INVALID; }
  for (int s = 0; s < b.ncol; ++s) {
    if (b.col_body[s] < 0 || b.col_body[s] >= b.nb || !(b.col_radius[s] >= 0) || !std::isfinite(b.col_radius[s])) { set_error("model: bad collision sphere"); return RSB_E_INVALID; }
    for (int a = 0; a < 3; ++a)
      if (!std::isfinite(b.col_pos[s][a])) { set_error("model: non-finite collision primitive position"); return RSB_E_INVALID; }
    if (!(b.col_rim[s] >= 0) || !std::isfinite(b.col_rim[s])) { set_error("model: col_rim must be finite and >= 0"); return RSB_E_INVALID; }
    if (b.col_rim[s] > 0) {
      const double n2 = b.col_axis[s][0] * b.col_axis[s][0] + b.col_axis[s][1] * b.col_axis[s][1] + b.col_axis[s][2] * b.col_axis[s][2];
      if (!(std::fabs(n2 - 1.0) < 1e-6)) { set_error("model: col_axis of a rim primitive must be a unit vector"); return RSB_E_INVALID; }
    }
    if (!std::memchr(b.col_material[s], 0, RSB_NAME_LEN)) { set_error("model: col_material is not NUL-terminated"); return RSB_E_INVALID; }
    if (b.col_capsule[s] == -1) {
      bool ok = s + 8 <= b.ncol;
      for (int e = 1; ok && e < 8; ++e) ok = b.col_body[s + e] == b.col_body[s] && b.col_capsule[s + e] == 0 && b.col_rim[s + e] == 0;
      if (!ok || b.col_rim[s] > 0) { set_error("model: col_capsule = -1 must head eight consecutive point primitives of one body (a box's corners)"); return RSB_E_INVALID; }
    } else if (b.col_capsule[s] != 0) {
      const int e = b.col_capsule[s] - 1;
      if (e < 0 || e >= b.ncol || e == s || b.col_body[e] != b.col_body[s] || b.col_radius[e] != b.col_radius[s] || b.col_rim[e] != b.col_rim[s] || b.col_capsule[e] != 0) {
        set_error("model: col_capsule must pair two sphere primitives (a capsule's ends) or two rim primitives (a cylinder's caps) of one body and one radius (set on the first only)"); return RSB_E_INVALID;
      }
    }
  }
  return RSB_OK;
}

}  // namespace rsb

// ---------------------------------------------------------------------------------- C ABI
extern "C" {

const char* rsb_last_error(void) { return rsb::last_error(); }
const char* rsb_version(void) { return "raisimlib_amd 0.1 (gfx950)"; }

int rsb_model_from_urdf_string(const char* xml, rsb_model** out) { return rsb_model_from_urdf_string_sampled(xml, 0.0, out); }
int rsb_model_from_urdf_file(const char* path, rsb_model** out) { return rsb_model_from_urdf_file_sampled(path, 0.0, out); }

int rsb_model_from_urdf_string_sampled(const char* xml, double sample_spacing, rsb_model** out) {
  if (!xml || !out || !(sample_spacing >= 0.0)) { rsb::set_error("rsb_model_from_urdf_string: null argument / negative spacing"); return RSB_E_INVALID; }
  try {
    auto m = std::make_unique<rsb_model>();
    rsb::build_from_xml(xml, &m->blob, &m->skipped_collisions, std::string(), sample_spacing);
    int st = rsb::validate_blob(m->blob);
    if (st != RSB_OK) return st;
    *out = m.release();
    return RSB_OK;
  } catch (const std::exception& e) {
    rsb::set_error(e.what());
    return RSB_E_PARSE;
  }
}

int rsb_model_from_urdf_file_sampled(const char* path, double sample_spacing, rsb_model** out) {
  if (!path || !out || !(sample_spacing >= 0.0)) { rsb::set_error("rsb_model_from_urdf_file: null argument / negative spacing"); return RSB_E_INVALID; }
  std::ifstream f(path);
  if (!f) { rsb::set_error(std::string("cannot open URDF file: ") + path); return RSB_E_INVALID; }
  std::stringstream ss;
  ss << f.rdbuf();
  if (!out) { rsb::set_error("rsb_model_from_urdf_file: null argument"); return RSB_E_INVALID; }
  try {
    auto m = std::make_unique<rsb_model>();
    std::string dir = path;
    const size_t cut = dir.find_last_of('/');
    dir = cut == std::string::npos ? std::string(".") : dir.substr(0, cut);
    rsb::build_from_xml(ss.str(), &m->blob, &m->skipped_collisions, dir, sample_spacing);   // mesh files are looked up relative to the URDF
    int st = rsb::validate_blob(m->blob);
    if (st != RSB_OK) return st;
    *out = m.release();
    return RSB_OK;
  } catch (const std::exception& e) {
    rsb::set_error(e.what());
    return RSB_E_PARSE;
  }
}

int rsb_model_from_blob(const rsb_model_blob* blob, rsb_model** out) {
  if (!blob || !out) { rsb::set_error("rsb_model_from_blob: null argument"); return RSB_E_INVALID; }
  int st = rsb::validate_blob(*blob);
  if (st != RSB_OK) return st;
  auto m = std::make_unique<rsb_model>();
  m->blob = *blob;
  m->skipped_collisions = 0;
  *out = m.release();
  return RSB_OK;
}

int rsb_model_destroy(rsb_model* m) { delete m; return RSB_OK; }

int rsb_model_get_blob(const rsb_model* m, rsb_model_blob* out) {
  if (!m || !out) { rsb::set_error("rsb_model_get_blob: null argument"); return RSB_E_INVALID; }
  *out = m->blob;
  return RSB_OK;
}

int rsb_model_body_index(const rsb_model* m, const char* link_name) {
  if (!m || !link_name) return RSB_E_INVALID;
  for (int i = 0; i < m->blob.nb; ++i) if (std::strcmp(m->blob.body_name[i], link_name) == 0) return i;
  return RSB_E_INVALID;
}

int rsb_model_joint_index(const rsb_model* m, const char* joint_name) {
  if (!m || !joint_name) return RSB_E_INVALID;
  for (int i = 0; i < m->blob.nb; ++i) if (std::strcmp(m->blob.joint_name[i], joint_name) == 0) return i;
  return RSB_E_INVALID;
}

double rsb_model_total_mass(const rsb_model* m) {
  double s = 0;
  if (m) for (int i = 0; i < m->blob.nb; ++i) s += m->blob.mass[i];
  return s;
}

int rsb_model_skipped_collisions(const rsb_model* m) { return m ? m->skipped_collisions : RSB_E_INVALID; }

const char* rsb_model_collision_material(const rsb_model* m, int collision) {
  if (!m || collision < 0 || collision >= m->blob.ncol) return nullptr;
  return m->blob.col_material[collision][0] ? m->blob.col_material[collision] : "default";
}

}  // extern "C"

// <mesh> colliders keep up to n support vertices of their convex hull (default 8, at most 26); applies to the models loaded afterwards
extern "C" int rsb_set_mesh_point_budget(int n) {
  if (n < 1 || n > rsb::kMeshPointsMax) { rsb::set_error("rsb_set_mesh_point_budget: 1 .. 26 points per mesh"); return RSB_E_INVALID; }
  rsb::g_mesh_points = n;
  return RSB_OK;
}
