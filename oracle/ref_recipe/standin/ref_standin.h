// ref_standin.h -- FUNCTIONAL stand-ins for the un-vendored upstream API that the reference's own hot-path sources are written
// against (Eigen, OpenCV's cv::Mat, glog, config_utilities, spatial_hash, spark_dsg, Hydra's containers).
//
// TEST INFRASTRUCTURE ONLY (like everything under oracle/).  Purpose: /root/reference holds the LOGIC of
//   khronos/src/active_window/integration/tracking_integrator.cpp        
//   khronos/src/active_window/motion_detection/free_space_motion_detector.cpp   (a9 - a11)
//   khronos/src/utils/geometry_utils.cpp                                 (a16, cluster bounding boxes)
//   khronos/src/active_window/object_detection/connected_semantics.cpp, instance_forwarding.cpp   (a18 / f3)
//   khronos/src/active_window/tracking/max_iou_tracker.cpp, external_tracker.cpp, data/track.cpp   (a18)
//   khronos/src/active_window/data/frame_data_buffer.cpp                 (a17)
//   khronos/src/backend/change_detection/ray_verificator.cpp, ray_change_detector.cpp,
//       background/ray_background_change_detector.cpp, objects/ray_object_change_detector.cpp, backend/change_state.cpp   (f4)
//   khronos/src/active_window/object_extraction/mesh_object_extractor.cpp, integration/object_integrator.cpp   (a13, a12)
//   khronos/src/active_window/active_window.cpp, object_extraction/object_worker_pool.cpp                      (a1 - a3, a15)
// but not the containers they run on.  oracle/ref_recipe/build_ref.sh compiles those files FROM WHERE THEY LIE
// (nothing is copied) against this header into oracle/_ref/libref_khronos.so, and tests/test_cpu_ref_pin.py runs the
// reference's own code beside oracle/oracle.cpp on the same seeded sequences.  What that pins: every decision those files
// take (update order, thresholds, early-outs, removal rule, seed / cluster / merge / filter / paint logic).  What it does NOT
// pin: the semantics of the stand-ins themselves -- each is the ASSUMPTIONS.md item named beside it ([A.n]); the projective
// integrator and the mesh integrator are not in /root/reference at all and stay unpinned.
//
// Only what those files (and the reference headers they include) use is provided; names and signatures follow the call
// sites in /root/reference (cited), the bodies are ours.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <condition_variable>
#include <iostream>
#include <list>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <sstream>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

// ------------------------------------------------------------------------------------------------------------------ Eigen
namespace Eigen {
constexpr int Dynamic = -1;

// fixed 3 x 1 column vector (Point, BlockIndex, VoxelIndex, GlobalIndex) -- the only fixed shape the three files touch
template <typename T, int R, int C>
class Matrix {
  static_assert(R == 3 && C == 1, "stand-in: fixed matrices are 3-vectors");

 public:
  using Scalar = T;
  Matrix() : v_{} {}
  Matrix(T x, T y, T z) : v_{x, y, z} {}
  T& x() { return v_[0]; }
  T& y() { return v_[1]; }
  T& z() { return v_[2]; }
  const T& x() const { return v_[0]; }
  const T& y() const { return v_[1]; }
  const T& z() const { return v_[2]; }
  T& operator[](size_t i) { return v_[i]; }
  const T& operator[](size_t i) const { return v_[i]; }
  T& operator()(size_t i) { return v_[i]; }
  const T& operator()(size_t i) const { return v_[i]; }
  Matrix operator+(const Matrix& o) const { return Matrix(v_[0] + o.v_[0], v_[1] + o.v_[1], v_[2] + o.v_[2]); }
  Matrix operator-(const Matrix& o) const { return Matrix(v_[0] - o.v_[0], v_[1] - o.v_[1], v_[2] - o.v_[2]); }
  Matrix& operator+=(const Matrix& o) { return *this = *this + o; }
  template <typename S>
  Matrix operator*(S s) const { return Matrix(v_[0] * s, v_[1] * s, v_[2] * s); }
  template <typename S>
  Matrix operator/(S s) const { return Matrix(v_[0] / s, v_[1] / s, v_[2] / s); }
  bool operator==(const Matrix& o) const { return v_ == o.v_; }
  bool operator!=(const Matrix& o) const { return !(v_ == o.v_); }
  T squaredNorm() const { return v_[0] * v_[0] + v_[1] * v_[1] + v_[2] * v_[2]; }
  // [A.7] dot = (x x' + y y') + z z', cross by the textbook formula, normalized = component-wise division by the norm
  // (ray_verificator.cpp:99-131,333-346)
  T dot(const Matrix& o) const { return v_[0] * o.v_[0] + v_[1] * o.v_[1] + v_[2] * o.v_[2]; }
  Matrix cross(const Matrix& o) const {
    return Matrix(v_[1] * o.v_[2] - v_[2] * o.v_[1], v_[2] * o.v_[0] - v_[0] * o.v_[2], v_[0] * o.v_[1] - v_[1] * o.v_[0]);
  }
  Matrix normalized() const {
    const T n = norm();
    return Matrix(v_[0] / n, v_[1] / n, v_[2] / n);
  }
  friend Matrix operator*(T s, const Matrix& a) { return Matrix(s * a.v_[0], s * a.v_[1], s * a.v_[2]); }
  Matrix& operator-=(const Matrix& o) { return *this = *this - o; }
  // [A, ASSUMPTIONS.md C.2] Eigen's norm() returns the matrix's own scalar type: for an integer vector the square root of the
  // integer squared norm, computed in double and truncated back (numext::sqrt).  free_space_motion_detector.cpp:349 takes the
  // norm of a difference of int64 voxel indices.
  T norm() const {
    if (std::is_integral<T>::value) return static_cast<T>(std::sqrt(static_cast<double>(squaredNorm())));
    return static_cast<T>(std::sqrt(squaredNorm()));
  }
  template <typename U>
  Matrix<U, 3, 1> cast() const { return Matrix<U, 3, 1>(static_cast<U>(v_[0]), static_cast<U>(v_[1]), static_cast<U>(v_[2])); }
  T maxCoeff() const { return std::max(v_[0], std::max(v_[1], v_[2])); }
  template <typename F>
  int format(const F&) const { return 0; }  // (log output only)
  static Matrix Zero() { return Matrix(T(0), T(0), T(0)); }
  static Matrix Constant(T c) { return Matrix(c, c, c); }

 private:
  std::array<T, 3> v_;
};

// dynamic matrices: MatrixXi (cluster overlap table, free_space_motion_detector.cpp:279-287) and the feature vector's
// Zero(1, 1) (measurement_clusters.h:52)
template <typename T>
class Matrix<T, Dynamic, Dynamic> {
 public:
  Matrix() = default;
  Matrix(size_t rows, size_t cols) : r_(rows), c_(cols), d_(rows * cols) {}
  void setZero() { std::fill(d_.begin(), d_.end(), T(0)); }
  T& operator()(size_t i, size_t j) { return d_[i * c_ + j]; }
  const T& operator()(size_t i, size_t j) const { return d_[i * c_ + j]; }
  int rows() const { return static_cast<int>(r_); }
  int cols() const { return static_cast<int>(c_); }
  static Matrix Zero(size_t rows, size_t cols) {
    Matrix m(rows, cols);
    m.setZero();
    return m;
  }
  // element count, element access and the two operations of the feature mean (track.cpp:49-70) and the cosine score
  size_t size() const { return d_.size(); }
  T& operator()(size_t i) { return d_[i]; }
  const T& operator()(size_t i) const { return d_[i]; }
  Matrix operator+(const Matrix& o) const {
    Matrix m(r_, c_);
    for (size_t i = 0; i < d_.size(); ++i) m.d_[i] = d_[i] + o.d_[i];
    return m;
  }
  friend Matrix operator*(T s, const Matrix& a) {
    Matrix m(a.r_, a.c_);
    for (size_t i = 0; i < a.d_.size(); ++i) m.d_[i] = s * a.d_[i];
    return m;
  }

 private:
  size_t r_ = 0, c_ = 0;
  std::vector<T> d_;
};

constexpr int StreamPrecision = 0, DontAlignCols = 0;
struct IOFormat {
  template <typename... A>
  IOFormat(A&&...) {}
};
using Vector3f = Matrix<float, 3, 1>;
using Vector3d = Matrix<double, 3, 1>;
using Vector3i = Matrix<int, 3, 1>;
using MatrixXi = Matrix<int, Dynamic, Dynamic>;
using MatrixXf = Matrix<float, Dynamic, Dynamic>;
using VectorXf = Matrix<float, Dynamic, Dynamic>;

// sensor pose: translation() (free_space_motion_detector.cpp:80) and pose * point (max_iou_tracker.cpp:585)
class Isometry3d {
 public:
  using Vec = Matrix<double, 3, 1>;
  Vec& translation() { return t_; }
  const Vec& translation() const { return t_; }
  double& linear(int r, int c) { return r_[3 * r + c]; }
  std::array<double, 9> rotation() const {
    std::array<double, 9> r;
    std::copy(r_, r_ + 9, r.begin());
    return r;
  }
  Vec operator*(const Vec& p) const {
    return Vec((r_[0] * p[0] + r_[1] * p[1]) + r_[2] * p[2] + t_[0], (r_[3] * p[0] + r_[4] * p[1]) + r_[5] * p[2] + t_[1],
               (r_[6] * p[0] + r_[7] * p[1]) + r_[8] * p[2] + t_[2]);
  }

 private:
  double r_[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  Vec t_;
};
}  // namespace Eigen

// ----------------------------------------------------------------------------------------------------------------- OpenCV
#define CV_32SC1 4  // (stand-in: the type code is the element size)
namespace cv {
template <typename T, int N>
struct Vec {
  T val[N];
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
using Vec3f = Vec<float, 3>;

// row-major image with untyped storage; at<T>(row, col) as the reference uses it (free_space_motion_detector.cpp:168,174,391)
struct Size {
  int width = 0, height = 0;
};

class Mat {
 public:
  int rows = 0, cols = 0;
  Mat() = default;
  Size size() const { return Size{cols, rows}; }
  static Mat zeros(const Size& s, int type) { return Mat(s.height, s.width, static_cast<size_t>(type)); }
  Mat(int r, int c, size_t elem_bytes) : rows(r), cols(c), eb_(elem_bytes), d_(std::make_shared<std::vector<uint8_t>>(static_cast<size_t>(r) * c * elem_bytes, 0)) {}
  template <typename T>
  T& at(int r, int c) { return reinterpret_cast<T*>(d_->data())[static_cast<size_t>(r) * cols + c]; }
  template <typename T>
  const T& at(int r, int c) const { return reinterpret_cast<const T*>(d_->data())[static_cast<size_t>(r) * cols + c]; }
  uint8_t* data() { return d_ ? d_->data() : nullptr; }
  const uint8_t* data() const { return d_ ? d_->data() : nullptr; }
  size_t elemSize() const { return eb_; }
  bool empty() const { return rows == 0 || cols == 0; }
  // setTo(value, mask) on a 32-bit integer image (connected_semantics.cpp:206: object_image.setTo(0, object_image == id))
  void setTo(int value, const Mat& mask) {
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c)
        if (mask.at<uint8_t>(r, c)) at<int>(r, c) = value;
  }

 private:
  size_t eb_ = 0;
  std::shared_ptr<std::vector<uint8_t>> d_;  // (cv::Mat copies share their pixels)
};
// element-wise comparison of a 32-bit integer image with a scalar: 8-bit mask, 255 where equal
inline Mat operator==(const Mat& m, int value) {
  Mat out(m.rows, m.cols, 1);
  for (int r = 0; r < m.rows; ++r)
    for (int c = 0; c < m.cols; ++c) out.at<uint8_t>(r, c) = m.at<int>(r, c) == value ? 255 : 0;
  return out;
}
}  // namespace cv

// ------------------------------------------------------------------------------------------------------------------- glog
namespace ref_standin {
struct NullStream {
  template <typename T>
  NullStream& operator<<(const T&) { return *this; }
};
}  // namespace ref_standin
#define CHECK(condition) ::ref_standin::NullStream()
#define LOG(severity) ::ref_standin::NullStream()
#define LOG_IF(severity, condition) ::ref_standin::NullStream()

// ------------------------------------------------------------------------------------------------------- config_utilities
// declare_config() bodies are compiled but never called by the harness (it fills the Config structs directly); checkValid
// hands the config through (tracking_integrator.cpp:69, free_space_motion_detector.cpp:71)
namespace config {
enum CheckMode { GT, GE, LT, LE, EQ, NE };
struct ThreadNumConversion {};
// the key names the reference's declare_config() functions announce are RECORDED (ref_config_keys in the harness lists them; the
// product's YAML loader is held to that list in tests/test_cpu_ref_pin.py)
inline std::vector<std::string>& recordedKeys() {
  static thread_local std::vector<std::string> keys;
  return keys;
}
inline std::vector<std::string>& recordedDefaults() {  // "key=value" of the arithmetic fields, as the Config holds them
  static thread_local std::vector<std::string> values;
  return values;
}
template <typename T>
void recordField(const T& value, const std::string& key) {
  recordedKeys().push_back(key);
  if constexpr (std::is_arithmetic<T>::value) {
    std::ostringstream os;
    os.precision(9);
    os << key << "=" << +value;
    recordedDefaults().push_back(os.str());
  }
}
inline void name(const std::string&) {}
template <typename T>
void field(T& value, const std::string& key, const std::string& = "") { recordField(value, key); }
template <typename Conversion, typename T>
void field(T& value, const std::string& key, const std::string& = "") { recordField(value, key); }
// the validity constraints declare_config() states are recorded too: "<name> <mode> <bound>", "<name> in a,b,c", "<name> range lo hi"
inline std::vector<std::string>& recordedChecks() {
  static thread_local std::vector<std::string> checks;
  return checks;
}
template <typename T, typename U>
void check(const T&, CheckMode mode, const U& bound, const std::string& field_name) {
  static const char* const names[] = {"GT", "GE", "LT", "LE", "EQ", "NE"};
  std::ostringstream os;
  os << field_name << " " << names[mode] << " " << +bound;
  recordedChecks().push_back(os.str());
}
template <typename T>
void checkIsOneOf(const T&, std::initializer_list<T> allowed, const std::string& field_name) {
  std::ostringstream os;
  os << field_name << " in ";
  bool first = true;
  for (const T& a : allowed) {
    os << (first ? "" : ",") << +a;
    first = false;
  }
  recordedChecks().push_back(os.str());
}
inline void checkCondition(bool, const std::string& message) { recordedChecks().push_back("condition " + message); }
template <typename T>
void checkInRange(const T&, const T& lo, const T& hi, const std::string& field_name) {
  std::ostringstream os;
  os << field_name << " range " << +lo << " " << +hi;
  recordedChecks().push_back(os.str());
}
template <typename E>
void enum_field(E&, const std::string& key, const std::vector<std::string>&) { recordedKeys().push_back(key); }
template <typename E>
void enum_field(E&, const std::string& key, std::initializer_list<const char*>) { recordedKeys().push_back(key); }
template <typename T>
const T& checkValid(const T& c) { return c; }
template <typename T>
bool isValid(const T&) { return true; }
template <typename Base, typename Derived, typename... Rest>
struct RegistrationWithConfig {
  explicit RegistrationWithConfig(const std::string&) {}
};
// a configured, optional sub-module (active_window.h:84-87): the harness installs the factory
template <typename T>
struct VirtualConfig {
  std::function<std::unique_ptr<T>()> factory;
  VirtualConfig() = default;
  template <typename Cfg, typename = decltype(std::declval<const Cfg&>().create())>
  VirtualConfig(const Cfg& c) : factory([c]() -> std::unique_ptr<T> { return c.create(); }) {}  // (a concrete sub-module's config)
  void setOptional() {}
  std::unique_ptr<T> create() const { return factory ? factory() : nullptr; }
  explicit operator bool() const { return static_cast<bool>(factory); }
};
template <typename B, typename C>
void base(C&) {}
template <typename C>
std::string toString(const C&) { return std::string(); }
}  // namespace config

// ----------------------------------------------------------------------------------------------------------- spatial_hash
namespace spatial_hash {
using Point = Eigen::Vector3f;
using BlockIndex = Eigen::Vector3i;
using VoxelIndex = Eigen::Vector3i;
using GlobalIndex = Eigen::Matrix<int64_t, 3, 1>;
using VoxelKey = std::pair<BlockIndex, VoxelIndex>;
using BlockIndices = std::vector<BlockIndex>;
using VoxelIndices = std::vector<VoxelIndex>;
using GlobalIndices = std::vector<GlobalIndex>;
using VoxelKeys = std::vector<VoxelKey>;

// (container-internal: iteration order of the reference's unordered containers is implementation-defined; the pin test
//  compares sets, ASSUMPTIONS.md C.1)
template <typename V>
struct Hash3 {
  size_t operator()(const V& v) const {
    return static_cast<size_t>(static_cast<uint64_t>(v[0]) * 73856093ull ^ static_cast<uint64_t>(v[1]) * 19349669ull ^ static_cast<uint64_t>(v[2]) * 83492791ull);
  }
};
template <typename T>
using IndexMap3 = std::unordered_map<Eigen::Vector3i, T, Hash3<Eigen::Vector3i>>;
using IndexSet3 = std::unordered_set<Eigen::Vector3i, Hash3<Eigen::Vector3i>>;
using GlobalIndexSet = std::unordered_set<GlobalIndex, Hash3<GlobalIndex>>;
template <typename T>
using GlobalIndexMap = std::unordered_map<GlobalIndex, T, Hash3<GlobalIndex>>;

// [A.7] index of the cell a point falls in: floor(p * inv) per axis (connected_semantics.cpp:136)
template <typename IndexT>
IndexT indexFromPoint(const Point& p, float inv) {
  using S = typename IndexT::Scalar;
  return IndexT(static_cast<S>(std::floor(p[0] * inv)), static_cast<S>(std::floor(p[1] * inv)), static_cast<S>(std::floor(p[2] * inv)));
}

// [A.1] floor division / non-negative remainder
inline VoxelKey keyFromGlobalIndex(const GlobalIndex& g, size_t voxels_per_side) {
  const int64_t n = static_cast<int64_t>(voxels_per_side);
  BlockIndex b;
  VoxelIndex v;
  for (int a = 0; a < 3; ++a) {
    int64_t q = g[a] / n, r = g[a] % n;
    if (r < 0) { r += n; --q; }
    b[a] = static_cast<int>(q);
    v[a] = static_cast<int>(r);
  }
  return {b, v};
}

// [A.1] neighbourhoods: 6 = faces, 18 = + edges, 26 = + corners, self excluded
inline const std::vector<std::array<int, 3>>& neighborOffsets(int connectivity) {
  static const std::vector<std::array<int, 3>> all = [] {
    std::vector<std::array<int, 3>> o;
    for (int order = 1; order <= 3; ++order)
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx)
            if (std::abs(dx) + std::abs(dy) + std::abs(dz) == order) o.push_back({dx, dy, dz});
    return o;
  }();
  static const std::vector<std::array<int, 3>> six(all.begin(), all.begin() + 6), eighteen(all.begin(), all.begin() + 18);
  return connectivity == 6 ? six : (connectivity == 18 ? eighteen : all);
}

// [A.7] regular grid: toIndex(p) = floor(p * (1 / voxel_size)), toPoint(i) = (float(i) + 0.5) * voxel_size
// (max_iou_tracker.cpp:485,555)
template <typename IndexT>
class Grid {
 public:
  explicit Grid(float voxel_size) : voxel_size_(voxel_size), voxel_size_inv_(1.f / voxel_size) {}
  IndexT toIndex(const Point& p) const { return indexFromPoint<IndexT>(p, voxel_size_inv_); }
  Point toPoint(const IndexT& i) const {
    return Point((static_cast<float>(i[0]) + 0.5f) * voxel_size_, (static_cast<float>(i[1]) + 0.5f) * voxel_size_, (static_cast<float>(i[2]) + 0.5f) * voxel_size_);
  }

 private:
  float voxel_size_, voxel_size_inv_;
};

class NeighborSearch {  // free_space_motion_detector.cpp:213,249
 public:
  explicit NeighborSearch(int connectivity) : conn_(connectivity) {}
  GlobalIndices neighborIndices(const GlobalIndex& g) const {
    GlobalIndices out;
    for (const auto& o : neighborOffsets(conn_)) out.emplace_back(g[0] + o[0], g[1] + o[1], g[2] + o[2]);
    return out;
  }

 protected:
  int conn_;
};

// one block of a layer: voxels in x-fastest linear order [A.1]
template <typename VoxelT>
struct Block {
  using Ptr = std::shared_ptr<Block>;
  using VoxelType = VoxelT;
  BlockIndex index;
  size_t voxels_per_side = 0;
  float voxel_size = 0.f, voxel_size_inv = 0.f, block_size = 0.f;
  std::vector<VoxelT> voxels;

  Block(const BlockIndex& idx, size_t vps, float vs) : index(idx), voxels_per_side(vps), voxel_size(vs), voxel_size_inv(1.f / vs), block_size(vs * static_cast<float>(vps)), voxels(vps * vps * vps) {}
  virtual ~Block() = default;
  size_t numVoxels() const { return voxels.size(); }
  size_t linear(const VoxelIndex& v) const { return static_cast<size_t>(v[0]) + voxels_per_side * (static_cast<size_t>(v[1]) + voxels_per_side * static_cast<size_t>(v[2])); }
  VoxelIndex fromLinear(size_t i) const {
    const int n = static_cast<int>(voxels_per_side);
    return VoxelIndex(static_cast<int>(i) % n, (static_cast<int>(i) / n) % n, static_cast<int>(i) / (n * n));
  }
  VoxelT& getVoxel(size_t i) { return voxels[i]; }
  const VoxelT& getVoxel(size_t i) const { return voxels[i]; }
  VoxelT& getVoxel(const VoxelIndex& v) { return voxels[linear(v)]; }
  const VoxelT& getVoxel(const VoxelIndex& v) const { return voxels[linear(v)]; }
  VoxelKey getVoxelKey(size_t i) const { return {index, fromLinear(i)}; }
  Point origin() const { return Point(static_cast<float>(index[0]) * block_size, static_cast<float>(index[1]) * block_size, static_cast<float>(index[2]) * block_size); }
  // [A.1] voxel of a point: floor((p - origin) * voxel_size_inv); may land on -1 / vps (free_space_motion_detector.cpp:192-196)
  VoxelIndex getVoxelIndex(const Point& p) const {
    const Point o = origin();
    return VoxelIndex(static_cast<int>(std::floor((p[0] - o[0]) * voxel_size_inv)), static_cast<int>(std::floor((p[1] - o[1]) * voxel_size_inv)),
                      static_cast<int>(std::floor((p[2] - o[2]) * voxel_size_inv)));
  }
  bool isValidVoxelIndex(const VoxelIndex& v) const {
    const int n = static_cast<int>(voxels_per_side);
    return v[0] >= 0 && v[1] >= 0 && v[2] >= 0 && v[0] < n && v[1] < n && v[2] < n;
  }
  GlobalIndex getGlobalVoxelIndex(const VoxelIndex& v) const {
    const int64_t n = static_cast<int64_t>(voxels_per_side);
    return GlobalIndex(index[0] * n + v[0], index[1] * n + v[1], index[2] * n + v[2]);
  }
};

template <typename BlockT>
class Layer {
 public:
  using BlockPtr = std::shared_ptr<BlockT>;
  using ConstBlockPtr = std::shared_ptr<const BlockT>;
  const float voxel_size;
  const size_t voxels_per_side;
  const float block_size, block_size_inv;

  Layer(float vs, size_t vps) : voxel_size(vs), voxels_per_side(vps), block_size(vs * static_cast<float>(vps)), block_size_inv(1.f / (vs * static_cast<float>(vps))) {}
  BlockPtr getBlockPtr(const BlockIndex& i) {
    auto it = blocks_.find(i);
    return it == blocks_.end() ? nullptr : it->second;
  }
  ConstBlockPtr getBlockPtr(const BlockIndex& i) const {
    auto it = blocks_.find(i);
    return it == blocks_.end() ? nullptr : it->second;
  }
  // [A.1] block of a point: floor(p * block_size_inv)
  BlockIndex blockIndexOf(const Point& p) const {
    return BlockIndex(static_cast<int>(std::floor(p[0] * block_size_inv)), static_cast<int>(std::floor(p[1] * block_size_inv)), static_cast<int>(std::floor(p[2] * block_size_inv)));
  }
  BlockIndex getBlockIndex(const Point& p) const { return blockIndexOf(p); }  // mesh_object_extractor.cpp:220-221
  const BlockT& getBlock(const BlockIndex& i) const { return *blocks_.at(i); }
  // block by block (mesh_object_extractor.cpp:246)
  struct iterator {
    typename IndexMap3<std::shared_ptr<BlockT>>::iterator it;
    BlockT& operator*() const { return *it->second; }
    iterator& operator++() { ++it; return *this; }
    bool operator!=(const iterator& o) const { return it != o.it; }
  };
  iterator begin() { return iterator{blocks_.begin()}; }
  iterator end() { return iterator{blocks_.end()}; }
  struct const_iterator {
    typename IndexMap3<std::shared_ptr<BlockT>>::const_iterator it;
    const BlockT& operator*() const { return *it->second; }
    const_iterator& operator++() { ++it; return *this; }
    bool operator!=(const const_iterator& o) const { return it != o.it; }
  };
  const_iterator begin() const { return const_iterator{blocks_.begin()}; }
  const_iterator end() const { return const_iterator{blocks_.end()}; }
  void insertBlock(const std::shared_ptr<BlockT>& b) { blocks_[b->index] = b; }
  BlockPtr getBlockPtr(const Point& p) { return getBlockPtr(blockIndexOf(p)); }
  ConstBlockPtr getBlockPtr(const Point& p) const { return getBlockPtr(blockIndexOf(p)); }
  BlockT& allocateBlock(const BlockIndex& i) {
    auto& b = blocks_[i];
    if (!b) b = std::make_shared<BlockT>(i, voxels_per_side, voxel_size);
    return *b;
  }
  void removeBlock(const BlockIndex& i) { blocks_.erase(i); }
  bool hasBlock(const BlockIndex& i) const { return blocks_.count(i) != 0; }
  size_t numBlocks() const { return blocks_.size(); }
  BlockIndices allocatedBlockIndices() const {
    BlockIndices out;
    for (const auto& kv : blocks_) out.push_back(kv.first);
    return out;
  }
  BlockIndices blockIndicesWithCondition(const std::function<bool(const BlockT&)>& cond) const {
    BlockIndices out;
    for (const auto& kv : blocks_)
      if (cond(*kv.second)) out.push_back(kv.first);
    return out;
  }

 private:
  IndexMap3<BlockPtr> blocks_;
};

// neighbours of a voxel across block borders (tracking_integrator.cpp:172,192)
class VoxelNeighborSearch : public NeighborSearch {
 public:
  template <typename LayerT>
  VoxelNeighborSearch(const LayerT& layer, int connectivity) : NeighborSearch(connectivity), vps_(static_cast<int>(layer.voxels_per_side)) {}
  VoxelKeys neighborKeys(const VoxelKey& key) const {
    VoxelKeys out;
    for (const auto& o : neighborOffsets(conn_)) {
      BlockIndex b = key.first;
      VoxelIndex v(key.second[0] + o[0], key.second[1] + o[1], key.second[2] + o[2]);
      for (int a = 0; a < 3; ++a) {
        if (v[a] < 0) { v[a] += vps_; --b[a]; }
        else if (v[a] >= vps_) { v[a] -= vps_; ++b[a]; }
      }
      out.emplace_back(b, v);
    }
    return out;
  }

 private:
  int vps_;
};
}  // namespace spatial_hash

// -------------------------------------------------------------------------------------------------------------- spark_dsg
namespace spark_dsg {
struct Color {
  uint8_t r = 0, g = 0, b = 0, a = 255;
  static Color gray() { return Color{128, 128, 128, 255}; }
};
namespace colormaps {
inline Color quality(float) { return Color(); }  // (visualisation only, mesh_object_extractor.cpp:259)
}  // namespace colormaps
// axis-aligned box of a point set (the type the reference builds from a cluster's pixels: free_space_motion_detector.cpp:396)
struct BoundingBox {
  struct PointAdaptor {
    virtual ~PointAdaptor() = default;
    virtual size_t size() const = 0;
    virtual Eigen::Vector3f get(size_t index) const = 0;
  };
  Eigen::Vector3f min, max;
  Eigen::Vector3f world_P_center, dimensions;  // [A.7] centre = (min + max) / 2, dimensions = max - min
  bool valid = false;
  BoundingBox() = default;
  explicit BoundingBox(const PointAdaptor& points) {
    for (size_t i = 0; i < points.size(); ++i) include(points.get(i));
    finish();
  }
  explicit BoundingBox(const std::vector<Eigen::Vector3f>& points) {
    for (const auto& p : points) include(p);
    finish();
  }
  // [A.7] box of given dimensions around a centre (mesh_object_extractor.cpp:170-171)
  BoundingBox(const Eigen::Vector3f& dims, const Eigen::Vector3f& center) : world_P_center(center), dimensions(dims), valid(true) {
    for (int a = 0; a < 3; ++a) {
      min[a] = center[a] - dims[a] * 0.5f;
      max[a] = center[a] + dims[a] * 0.5f;
    }
  }
  // [A.7] union with another box; an invalid box takes the other one over (mesh_object_extractor.cpp:337)
  void merge(const BoundingBox& o) {
    if (!o.valid) return;
    include(o.min);
    include(o.max);
    finish();
  }
  float volume() const { return valid ? dimensions[0] * dimensions[1] * dimensions[2] : 0.f; }
  // [A.7] IoU of two axis-aligned boxes: intersection volume / union volume, 0 when disjoint or either box is invalid
  // (max_iou_tracker.cpp:592)
  float computeIoU(const BoundingBox& o) const {
    if (!valid || !o.valid) return 0.f;
    float inter = 1.f;
    for (int a = 0; a < 3; ++a) {
      const float lo = std::max(min[a], o.min[a]), hi = std::min(max[a], o.max[a]);
      if (hi <= lo) return 0.f;
      inter *= hi - lo;
    }
    return inter / (volume() + o.volume() - inter);
  }
  void include(const Eigen::Vector3f& p) {
    if (!valid) { min = max = p; valid = true; return; }
    for (int a = 0; a < 3; ++a) {
      if (p[a] < min[a]) min[a] = p[a];
      if (p[a] > max[a]) max[a] = p[a];
    }
  }
  void finish() {
    if (!valid) return;
    for (int a = 0; a < 3; ++a) {
      world_P_center[a] = (min[a] + max[a]) * 0.5f;
      dimensions[a] = max[a] - min[a];
    }
  }
};
using LayerId = int64_t;
using NodeId = uint64_t;
struct NodeSymbol {};

// mesh as geometry_utils.cpp:61-86 and ray_verificator.cpp:216-223 use it
struct Mesh {
  using Pos = Eigen::Vector3f;
  using Face = std::array<size_t, 3>;
  std::vector<Pos> points;
  std::vector<Color> colors;
  std::vector<uint32_t> labels;
  std::vector<uint64_t> first_seen_stamps, stamps;
  std::vector<Face> faces;
  size_t numVertices() const { return points.size(); }
  const Pos& pos(size_t i) const { return points.at(i); }
  uint64_t timestamp(size_t i) const { return stamps.at(i); }  // [A] the vertex's last-seen stamp (ray_background_change_detector.cpp:93)
};

// node attributes: the fields the change detection reads (ray_verificator.cpp:205-207,361-365; ray_verificator.h:170-174)
struct NodeAttributes {
  using Ptr = std::unique_ptr<NodeAttributes>;
  virtual ~NodeAttributes() = default;
  Eigen::Vector3d position;
};
struct AgentNodeAttributes : NodeAttributes {
  std::chrono::nanoseconds timestamp{0};
};
struct KhronosObjectAttributes : NodeAttributes {
  using Ptr = std::unique_ptr<KhronosObjectAttributes>;
  Mesh mesh;
  BoundingBox bounding_box;
  int semantic_label = -1;
  Eigen::VectorXf semantic_feature;
  std::vector<uint64_t> first_observed_ns, last_observed_ns;
  std::vector<Eigen::Vector3f> trajectory_positions;
  std::vector<uint64_t> trajectory_timestamps;
  std::vector<std::vector<Eigen::Vector3f>> dynamic_object_points;
  std::map<std::string, std::vector<size_t>> details;
};
struct SceneGraphNode {
  std::unique_ptr<NodeAttributes> attrs;
  template <typename T>
  T& attributes() const { return dynamic_cast<T&>(*attrs); }
};
struct SceneGraphLayer {
  using Nodes = std::map<NodeId, std::unique_ptr<SceneGraphNode>>;
  Nodes nodes_;
  const Nodes& nodes() const { return nodes_; }
};
struct DsgLayers {
  inline static const std::string AGENTS = "AGENTS";
  inline static const std::string OBJECTS = "OBJECTS";
};
// the graph: named layers, the agents' layer additionally keyed by the robot prefix (ray_verificator.cpp:187-190), one mesh
class DynamicSceneGraph {
 public:
  struct LayerKey {
    LayerId layer;
  };
  std::map<std::string, LayerId> layer_ids{{DsgLayers::OBJECTS, 2}, {DsgLayers::AGENTS, 3}};
  std::map<std::pair<LayerId, char>, SceneGraphLayer> layers;
  std::shared_ptr<Mesh> mesh_;
  std::optional<LayerKey> getLayerKey(const std::string& name) const {
    const auto it = layer_ids.find(name);
    return it == layer_ids.end() ? std::nullopt : std::optional<LayerKey>(LayerKey{it->second});
  }
  const SceneGraphLayer* findLayer(LayerId layer, char prefix = 0) const {
    const auto it = layers.find({layer, prefix});
    return it == layers.end() ? nullptr : &it->second;
  }
  const SceneGraphLayer& getLayer(LayerId layer, char prefix = 0) const { return layers.at({layer, prefix}); }
  const SceneGraphLayer& getLayer(const std::string& name) const { return layers.at({layer_ids.at(name), 0}); }
  bool hasLayer(const std::string& name) const { return layer_ids.count(name) && layers.count({layer_ids.at(name), 0}); }
  bool hasMesh() const { return mesh_ != nullptr; }
  std::shared_ptr<Mesh> mesh() const { return mesh_; }
};
}  // namespace spark_dsg

// ------------------------------------------------------------------------------------------------------------------ Hydra
namespace hydra {
using TimeStamp = uint64_t;
// [A.6] nanoseconds -> double seconds (the reference compares stamps in seconds: tracking_integrator.cpp:237-238,250)
inline double toSeconds(TimeStamp ns) { return static_cast<double>(ns) / 1e9; }
inline TimeStamp fromSeconds(double s) { return static_cast<TimeStamp>(s * 1e9); }

using Point = spatial_hash::Point;
using BlockIndex = spatial_hash::BlockIndex;
using VoxelIndex = spatial_hash::VoxelIndex;
using GlobalIndex = spatial_hash::GlobalIndex;
using VoxelKey = spatial_hash::VoxelKey;
using BlockIndices = spatial_hash::BlockIndices;
using VoxelIndices = spatial_hash::VoxelIndices;
using GlobalIndices = spatial_hash::GlobalIndices;
using VoxelKeys = spatial_hash::VoxelKeys;
using BlockIndexSet = spatial_hash::IndexSet3;
using VoxelIndexSet = spatial_hash::IndexSet3;
using GlobalIndexSet = spatial_hash::GlobalIndexSet;
template <typename T>
using BlockIndexMap = spatial_hash::IndexMap3<T>;
template <typename T>
using VoxelIndexMap = spatial_hash::IndexMap3<T>;
template <typename T>
using GlobalIndexMap = spatial_hash::GlobalIndexMap<T>;

using FeatureVector = Eigen::VectorXf;
using FeatureMap = std::unordered_map<int, FeatureVector>;

// [A.6] voxel types: the fields the reference reads and writes
struct TsdfVoxel {
  float distance = 0.f;
  float weight = 0.f;
  spark_dsg::Color color;
};
struct TrackingVoxel {
  TimeStamp last_observed = 0u;
  TimeStamp last_occupied = 0u;
  bool active = false;
  bool ever_free = false;
  bool to_remove = false;
};
// [A.4] binary confidence voxel: two counters and the "never updated" flag (mesh_object_extractor.cpp:342-356)
struct SemanticVoxel {
  bool empty = true;
  Eigen::VectorXf semantic_likelihoods = Eigen::VectorXf::Zero(2, 1);
};

struct TsdfBlock : spatial_hash::Block<TsdfVoxel> {
  using Ptr = std::shared_ptr<TsdfBlock>;
  using spatial_hash::Block<TsdfVoxel>::Block;
  mutable bool updated = false, mesh_updated = false, tracking_updated = false;
  static bool trackingUpdated(const TsdfBlock& b) { return b.tracking_updated; }  // tracking_integrator.cpp:77
  void clearUpdated() const { updated = false; }  // [A.6] clears `updated` only (active_window.cpp:169-171)
};
struct TrackingBlock : spatial_hash::Block<TrackingVoxel> {
  using Ptr = std::shared_ptr<TrackingBlock>;
  using spatial_hash::Block<TrackingVoxel>::Block;
  bool has_active_data = false;
};
struct SemanticBlock : spatial_hash::Block<SemanticVoxel> {
  using spatial_hash::Block<SemanticVoxel>::Block;
};
using TsdfLayer = spatial_hash::Layer<TsdfBlock>;
using TrackingLayer = spatial_hash::Layer<TrackingBlock>;
using SemanticLayer = spatial_hash::Layer<SemanticBlock>;

using Mesh = spark_dsg::Mesh;
using MeshBlock = Mesh;
using MeshLayer = std::vector<MeshBlock>;  // (iterated block by block, geometry_utils.cpp:64)

// the map: TSDF layer always, tracking layer optional (mesh_object_extractor.cpp:208-211); removeBlock drops a block from every layer
class VolumetricMap {
 public:
  struct Config {
    float voxel_size = 0.1f;
    float truncation_distance = 0.3f;
    size_t voxels_per_side = 16;
    bool with_semantics = false;
    bool with_tracking = true;
  } const config;
  explicit VolumetricMap(const Config& c)
      : config(c), tsdf_(c.voxel_size, c.voxels_per_side), tracking_(std::make_shared<TrackingLayer>(c.voxel_size, c.voxels_per_side)),
        semantic_(std::make_shared<SemanticLayer>(c.voxel_size, c.voxels_per_side)) {}
  std::shared_ptr<SemanticLayer> getSemanticLayer() { return semantic_; }
  MeshLayer& getMeshLayer() { return mesh_; }
  const MeshLayer& getMeshLayer() const { return mesh_; }
  std::shared_ptr<void> backend;  // (the harness's CPU-oracle map behind this one, where the integrators are bridged)
  std::function<void(const BlockIndex&)> on_remove;  // (bridge: the same block leaves the backend)
  bool hasSemantics() const { return config.with_semantics; }
  // [A.6] deep copy of the blocks whose `updated` flag is set (active_window.cpp:229)
  std::shared_ptr<VolumetricMap> cloneUpdated() const {
    auto out = std::make_shared<VolumetricMap>(config);
    for (const TsdfBlock& b : tsdf_) {
      if (!b.updated) continue;
      out->tsdf_.insertBlock(std::make_shared<TsdfBlock>(b));
      if (auto t = tracking_->getBlockPtr(b.index)) out->tracking_->insertBlock(std::make_shared<TrackingBlock>(*t));
    }
    return out;
  }
  TsdfLayer& getTsdfLayer() { return tsdf_; }
  const TsdfLayer& getTsdfLayer() const { return tsdf_; }
  std::shared_ptr<TrackingLayer> getTrackingLayer() { return tracking_; }
  std::shared_ptr<const TrackingLayer> getTrackingLayer() const { return tracking_; }
  void allocateBlock(const BlockIndex& i) {
    tsdf_.allocateBlock(i);
    if (config.with_tracking) tracking_->allocateBlock(i);
    if (config.with_semantics) semantic_->allocateBlock(i);
  }
  void removeBlock(const BlockIndex& i) {
    tsdf_.removeBlock(i);
    tracking_->removeBlock(i);
    semantic_->removeBlock(i);
    if (on_remove) on_remove(i);
  }

 private:
  TsdfLayer tsdf_;
  std::shared_ptr<TrackingLayer> tracking_;
  std::shared_ptr<SemanticLayer> semantic_;
  MeshLayer mesh_;
};

// ---- the two integrators whose arithmetic is NOT in /root/reference.  The reference's object extractor drives them
// (mesh_object_extractor.cpp:238-243,267); here a call is handed to whatever the harness installed -- the CPU oracle, i.e. the
// ASSUMPTIONS.md A.3 / A.5 semantics -- so that the reference's own glue around them can run.
class ProjectiveIntegrator;
class MeshIntegrator;
struct InputData;
}  // namespace hydra
namespace ref_standin {
struct Bridge {
  std::function<void(const hydra::ProjectiveIntegrator&, const hydra::InputData&, hydra::VolumetricMap&, bool, const cv::Mat&)> integrate;
  std::function<void(const hydra::MeshIntegrator&, hydra::VolumetricMap&, bool, bool)> mesh;
  std::function<void(hydra::InputData&)> parse_input;  // range image + world-frame vertex map of a raw frame (ASSUMPTIONS.md A.2)
};
inline Bridge& bridge() {
  static Bridge b;
  return b;
}
}  // namespace ref_standin
namespace hydra {
struct BinarySemanticIntegrator {
  struct Config {};
};
struct InterpolationWeights {
  float w[4] = {0, 0, 0, 0};
  int u[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
};
struct VoxelMeasurement {
  float sdf = 0.f;
  InterpolationWeights interpolation_weights;
  int label = -1;
};
// [A.3] interpolateID = the value at the max-weight pixel (first maximum) (object_integrator.cpp:70,77)
struct Interpolator {
  int interpolateID(const cv::Mat& image, const InterpolationWeights& w) const {
    int best = 0;
    for (int k = 1; k < 4; ++k)
      if (w.w[k] > w.w[best]) best = k;
    return image.at<int>(w.v[best], w.u[best]);
  }
};
class ProjectiveIntegrator {
 public:
  struct Config {
    struct SemanticIntegratorConfig {
      bool binary = false;
      SemanticIntegratorConfig& operator=(const BinarySemanticIntegrator::Config&) {
        binary = true;
        return *this;
      }
    } semantic_integrator;
  };
  using VoxelMeasurement = hydra::VoxelMeasurement;  // (named unqualified inside the subclass, object_integrator.h:69)
  explicit ProjectiveIntegrator(const Config& c) : config(c), interpolator_(std::make_unique<Interpolator>()) {}
  virtual ~ProjectiveIntegrator() = default;
  void updateMap(const InputData& data, VolumetricMap& map, bool allocate_blocks = true, const cv::Mat& integration_mask = cv::Mat()) const {
    ref_standin::bridge().integrate(*this, data, map, allocate_blocks, integration_mask);
  }
  const Config config;

 protected:
  virtual bool computeLabel(const VolumetricMap::Config&, const InputData&, const cv::Mat&, VoxelMeasurement&) const { return true; }
  std::unique_ptr<Interpolator> interpolator_;
};
struct MeshIntegratorConfig {};
class MeshIntegrator {
 public:
  explicit MeshIntegrator(const MeshIntegratorConfig&) {}
  void generateMesh(VolumetricMap& map, bool only_mesh_updated_blocks, bool clear_updated_flag) const {
    ref_standin::bridge().mesh(*this, map, only_mesh_updated_blocks, clear_updated_flag);
  }
};

// thread-safe dispenser of a block list (tracking_integrator.cpp:83-86,140)
template <typename IndexT>
class IndexGetter {
 public:
  explicit IndexGetter(const std::vector<IndexT>& indices) : indices_(indices) {}
  bool getNextIndex(IndexT& out) {
    const size_t i = next_.fetch_add(1);
    if (i >= indices_.size()) return false;
    out = indices_[i];
    return true;
  }

 private:
  std::vector<IndexT> indices_;
  std::atomic<size_t> next_{0};
};

class GlobalInfo {
 public:
  struct Config {
    int default_verbosity = 0;
    int default_num_threads = 2;
    bool store_visualization_details = false;
  };
  struct LabelNames {
    std::string at(int) const { return "label"; }  // (log output only, mesh_object_extractor.cpp:372)
  };
  LabelNames getLabelToNameMap() const { return {}; }
  static GlobalInfo& instance() {
    static GlobalInfo g;
    return g;
  }
  const Config& getConfig() const { return config_; }
  // [A.7] isObject(label) = membership in the configured object-label list (connected_semantics.cpp:133,153)
  struct LabelSpaceConfig {
    std::unordered_set<int> object_labels;
    bool isObject(int label) const { return object_labels.count(label) != 0; }
  };
  const LabelSpaceConfig& getLabelSpaceConfig() const { return labels_; }
  LabelSpaceConfig& mutableLabelSpaceConfig() { return labels_; }  // (harness only)

 private:
  Config config_;
  LabelSpaceConfig labels_;
};

// [A.8] pinhole projection to the nearest pixel; false when behind the camera or outside the image (max_iou_tracker.cpp:586)
// (file import of change states, change_state.cpp:70-145: not exercised by the harness)
struct CsvReader {
  explicit CsvReader(const std::string&) {}
  bool isSetup() const { return false; }
  bool hasHeader(const std::string&) const { return false; }
  bool checkRequiredHeaders(const std::vector<std::string>&) const { return false; }
  size_t numRows() const { return 0; }
  std::string getEntry(const std::string&, size_t) const { return "0"; }
};

struct RobotPrefixConfig {
  char key = 'a';
};

struct Sensor {
  int width = 0, height = 0;
  float fx = 1.f, fy = 1.f, cx = 0.f, cy = 0.f;
  bool projectPointToImagePlane(const Eigen::Vector3f& p, int& u, int& v) const {
    if (p[2] <= 0.f) return false;
    u = static_cast<int>(std::round((p[0] * fx) / p[2] + cx));
    v = static_cast<int>(std::round((p[1] * fy) / p[2] + cy));
    return u >= 0 && v >= 0 && u < width && v < height;
  }
};

// [A.7] cosine score a . b / (|a| |b|) (max_iou_tracker.cpp:56-59)
struct EmbeddingDistance {
  virtual ~EmbeddingDistance() = default;
  virtual float score(const FeatureVector& a, const FeatureVector& b) const = 0;
};
struct CosineDistance : EmbeddingDistance {
  struct Config {
    std::unique_ptr<EmbeddingDistance> create() const { return std::make_unique<CosineDistance>(); }
  };
  float score(const FeatureVector& a, const FeatureVector& b) const override {
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (size_t i = 0; i < a.size(); ++i) {
      ab += a(i) * b(i);
      aa += a(i) * a(i);
      bb += b(i) * b(i);
    }
    return ab / (std::sqrt(aa) * std::sqrt(bb));
  }
};

// [A.9] a set of prompt embeddings; the best score of a feature against it (instance_forwarding.cpp:97-101)
struct EmbeddingGroup {
  using Ptr = std::unique_ptr<EmbeddingGroup>;
  struct ScoreResult {
    float score = 0.f;
    size_t index = 0;
  };
  std::vector<FeatureVector> embeddings;
  ScoreResult getBestScore(const EmbeddingDistance& metric, const FeatureVector& feature) const {
    ScoreResult best;
    best.score = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < embeddings.size(); ++i) {
      const float sc = metric.score(embeddings[i], feature);
      if (sc > best.score) {
        best.score = sc;
        best.index = i;
      }
    }
    return best;
  }
};


// the input of a frame: what the motion detector reads (free_space_motion_detector.cpp:74,80,114,168,174)
struct InputData {
  using RangeType = float;
  using VertexType = cv::Vec3f;
  using LabelType = int;
  TimeStamp timestamp_ns = 0;
  cv::Mat label_image;
  std::map<int, FeatureVector> label_features;  // open-set: one feature per instance id (instance_forwarding.cpp:96,140)
  cv::Mat vertex_map;   // world frame
  cv::Mat range_image;
  Eigen::Isometry3d world_T_sensor;
  const Eigen::Isometry3d& getSensorPose() const { return world_T_sensor; }
  Sensor sensor;
  const Sensor& getSensor() const { return sensor; }
  Eigen::Isometry3d world_T_body;
  cv::Mat depth_image;
  // (bridge) the raw frame, for the integrator behind ref_standin::bridge()
  std::shared_ptr<std::vector<float>> depth;
  std::shared_ptr<std::vector<uint8_t>> rgb;
  double world_T_sensor16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

namespace timing {
struct ScopedTimer {
  ScopedTimer(const std::string&, TimeStamp) {}
  void stop() {}
};
struct ElapsedTimeRecorder {
  static ElapsedTimeRecorder& instance() {
    static ElapsedTimeRecorder r;
    return r;
  }
  template <typename D>
  void record(const std::string&, TimeStamp, const D&) {}
};
}  // namespace timing

// thread-safe queue (object_worker_pool.cpp:96,121-131)
template <typename T>
class MessageQueue {
 public:
  using Ptr = std::shared_ptr<MessageQueue>;
  void push(const T& v) {
    {
      std::lock_guard<std::mutex> lock(m_);
      q_.push_back(v);
    }
    cv_.notify_all();
  }
  bool poll(size_t wait_us) {
    std::unique_lock<std::mutex> lock(m_);
    return cv_.wait_for(lock, std::chrono::microseconds(wait_us), [this] { return !q_.empty(); });
  }
  T pop() {
    std::lock_guard<std::mutex> lock(m_);
    T v = q_.front();
    q_.pop_front();
    return v;
  }
  size_t size() const {
    std::lock_guard<std::mutex> lock(m_);
    return q_.size();
  }

 private:
  mutable std::mutex m_;
  std::condition_variable cv_;
  std::list<T> q_;
};

// output sinks (active_window.h:69, active_window.cpp:80,146)
template <typename... Args>
struct OutputSink {
  using Ptr = std::shared_ptr<OutputSink>;
  using List = std::list<Ptr>;
  using Factory = std::function<Ptr()>;
  virtual ~OutputSink() = default;
  virtual void call(Args...) const {}
  static List instantiate(const std::vector<Factory>& factories) {
    List l;
    for (const auto& f : factories)
      if (auto s = f()) l.push_back(s);
    return l;
  }
  static void callAll(const List& sinks, Args... args) {
    for (const auto& s : sinks) s->call(args...);
  }
  static std::string printSinks(const List&) { return std::string(); }
};

// a raw frame as it reaches the module, and what the module hands on (active_window.cpp:118-174,217-249)
struct InputPacket {
  TimeStamp timestamp_ns = 0;
  int width = 0, height = 0;
  Sensor sensor;
  std::shared_ptr<std::vector<float>> depth;
  std::shared_ptr<std::vector<uint8_t>> rgb;
  std::shared_ptr<std::vector<int32_t>> label;
  double world_T_sensor16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};
struct LayerUpdate {
  using Ptr = std::shared_ptr<LayerUpdate>;
  explicit LayerUpdate(spark_dsg::LayerId l) : layer(l) {}
  spark_dsg::LayerId layer;
  std::list<spark_dsg::NodeAttributes::Ptr> attributes;
};
struct ActiveWindowOutput {
  using Ptr = std::shared_ptr<ActiveWindowOutput>;
  TimeStamp timestamp_ns = 0;
  Eigen::Vector3d world_t_body;
  std::array<double, 9> world_R_body{};
  std::shared_ptr<VolumetricMap> map;
  void setMap(const std::shared_ptr<VolumetricMap>& m) { map = m; }
  BlockIndices archived_mesh_indices;
  std::map<spark_dsg::LayerId, LayerUpdate::Ptr> graph_update;
  std::shared_ptr<InputData> sensor_data;
};
class ActiveWindowModule {
 public:
  struct Config {
    VolumetricMap::Config volumetric_map;
    Config(bool with_semantics, bool with_tracking) {
      volumetric_map.with_semantics = with_semantics;
      volumetric_map.with_tracking = with_tracking;
    }
  };
  using OutputQueue = MessageQueue<ActiveWindowOutput::Ptr>;
  using InputQueue = MessageQueue<std::shared_ptr<InputPacket>>;
  ActiveWindowModule(const Config& c, const OutputQueue::Ptr& out)
      : map_(c.volumetric_map), input_queue_(std::make_shared<InputQueue>()), output_queue_(out ? out : std::make_shared<OutputQueue>()) {}
  virtual ~ActiveWindowModule() = default;
  virtual std::string printInfo() const { return std::string(); }

 protected:
  virtual ActiveWindowOutput::Ptr spinOnce(const InputPacket& input) = 0;
  VolumetricMap map_;
  std::shared_ptr<InputQueue> input_queue_;
  OutputQueue::Ptr output_queue_;
};

namespace conversions {
// [A.2] the input conversion: depth, labels, colour taken over; range image + world-frame vertex map through the bridge
inline std::unique_ptr<InputData> parseInputPacket(const InputPacket& in, bool, bool) {
  auto d = std::make_unique<InputData>();
  d->timestamp_ns = in.timestamp_ns;
  d->sensor = in.sensor;
  d->depth = in.depth;
  d->rgb = in.rgb;
  std::copy(in.world_T_sensor16, in.world_T_sensor16 + 16, d->world_T_sensor16);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) d->world_T_sensor.linear(r, c) = in.world_T_sensor16[4 * r + c];
    d->world_T_sensor.translation()[r] = in.world_T_sensor16[4 * r + 3];
  }
  d->world_T_body = d->world_T_sensor;
  d->depth_image = cv::Mat(in.height, in.width, sizeof(float));
  std::copy(in.depth->begin(), in.depth->end(), reinterpret_cast<float*>(d->depth_image.data()));
  d->label_image = cv::Mat(in.height, in.width, sizeof(int));
  if (in.label) std::copy(in.label->begin(), in.label->end(), reinterpret_cast<int*>(d->label_image.data()));
  d->range_image = cv::Mat(in.height, in.width, sizeof(float));
  d->vertex_map = cv::Mat(in.height, in.width, sizeof(cv::Vec3f));
  ref_standin::bridge().parse_input(*d);
  return d;
}
}  // namespace conversions

// [A.3] mask of the non-zero pixels of an id image (active_window.cpp:209)
inline void maskNonZero(const cv::Mat& in, cv::Mat& out) {
  out = cv::Mat(in.rows, in.cols, sizeof(int));
  for (int r = 0; r < in.rows; ++r)
    for (int c = 0; c < in.cols; ++c) out.at<int>(r, c) = in.at<int>(r, c) != 0 ? 1 : 0;
}
}  // namespace hydra
