// Engine3D.hpp -- C++17 host-side mirror of the reference's public raytrace API over the C ABI of
// include/softray.h.  The reference is compiled code (C#, .NET Framework 4.0) whose toolchain is absent
// here, so the host layer above the boundary is written in C++ with the reference's names, argument
// meaning and error behaviour:
//
//   Engine3D::Vector / Color                      Engine3D/Vector.cs, Engine3D/Color.cs
//   Engine3D::Raytrace::Sphere / Plane / Triangle / GeometryCollection
//                                                 Engine3D/Raytrace/{Sphere,Plane,Triangle,GeometryCollection}.cs
//   Engine3D::Model / Instance                    Engine3D/Model.cs, Engine3D/Instance.cs
//   Engine3D::Renderer                            Engine3D/Renderer.cs (raytrace half: :35-139, :593, :629, :673, :701)
//
// Header-only; link with -lsoftray_hip.  Nothing is traced on the host: Render() fills an sr_frame the way
// Renderer.RaytraceGeometry does (Renderer.cs:1501-1687) and calls sr_render.  Features outside the hot
// path (rasteriser, static shadows, AO, light field, path tracing, voxels) throw std::logic_error.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <istream>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../../include/softray.h"

namespace Engine3D {

struct ArgumentOutOfRangeException : std::out_of_range { using std::out_of_range::out_of_range; };   // SpatialSubdivision.cs:293
struct FormatException : std::runtime_error { using std::runtime_error::runtime_error; };           // Model.cs:555, ThreeDSFile.cs:168
struct InvalidOperationException : std::runtime_error { using std::runtime_error::runtime_error; };

inline void sr_check(int rc) {
    if (rc == SR_OK) return;
    std::string msg = sr_last_error();
    switch (rc) {
        case SR_ERR_OUT_OF_RANGE: throw ArgumentOutOfRangeException(msg);
        case SR_ERR_FORMAT: throw FormatException(msg);
        case SR_ERR_INVALID_ARG: throw std::invalid_argument(msg);
        default: throw InvalidOperationException(msg);
    }
}

struct Vector {                                            // Vector.cs:9-31
    double x = 0, y = 0, z = 0;
    Vector() = default;
    Vector(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    Vector operator-(const Vector& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vector operator*(double s) const { return {x * s, y * s, z * s}; }
    void Normalise() {                                     // Vector.cs:177-185
        double inv = 1.0 / std::sqrt(x * x + y * y + z * z);
        x *= inv; y *= inv; z *= inv;
    }
};

inline uint8_t ToByte(double d) {                          // C# unchecked (byte)(double)
    if (!(d > -2147483649.0 && d < 2147483648.0)) return 0;
    return (uint8_t)(int32_t)d;
}

struct Color {                                             // Color.cs:5-36
    double r = 0, g = 0, b = 0;
    uint32_t ToARGB() const {                              // Color.cs:105-111
        return (255u << 24) + ((uint32_t)ToByte(r * 255.0) << 16) + ((uint32_t)ToByte(g * 255.0) << 8) + ToByte(b * 255.0);
    }
    static Color White() { return {1, 1, 1}; }  static Color Red() { return {1, 0, 0}; }
    static Color Green() { return {0, 1, 0}; }  static Color Blue() { return {0, 0, 1}; }
    static Color Yellow() { return {1, 1, 0}; } static Color Cyan() { return {0, 1, 1}; }
};

namespace Raytrace {

struct Sphere {                                            // Sphere.cs:26-33
    Vector center; double radius; Engine3D::Color Color = Engine3D::Color::White();
    Sphere(Vector c, double r) : center(c), radius(r) { if (!(r > 0)) throw std::invalid_argument("radius > 0"); }
};
struct Plane {                                             // Plane.cs:22-29
    Vector point, normal; Engine3D::Color Color = Engine3D::Color::White();
    Plane(Vector p, Vector n) : point(p), normal(n) {}
};
struct Triangle {                                          // Triangle.cs:29-57
    Vector v1, v2, v3; uint32_t color;
    Triangle(Vector a, Vector b, Vector c, uint32_t col) : v1(a), v2(b), v3(c), color(col) {}
};

struct AxisAlignedBox {                                    // AxisAlignedBox.cs:15-28 as an IRayIntersectable (IntersectRay :60-95)
    Vector Min, Max;
    AxisAlignedBox(Vector mn, Vector mx) : Min(mn), Max(mx) {
        if (!(mn.x < mx.x && mn.y < mx.y && mn.z < mx.z)) throw std::invalid_argument("Axis aligned bounding box has bad coordinates");
    }
};

class GeometryCollection {                                 // GeometryCollection.cs:8-31
public:
    using Item = std::variant<Sphere, Plane, Triangle, AxisAlignedBox>;
    void Add(const Item& g) { items_.push_back(g); }
    int Count() const { return (int)items_.size(); }
    std::vector<sr_prim> ToPrims() const {
        std::vector<sr_prim> out;
        for (const Item& it : items_) {
            sr_prim p{};
            if (auto s = std::get_if<Sphere>(&it)) {
                p.kind = 0; p.argb = s->Color.ToARGB();
                p.p[0] = s->center.x; p.p[1] = s->center.y; p.p[2] = s->center.z; p.p[3] = s->radius;
            } else if (auto pl = std::get_if<Plane>(&it)) {
                p.kind = 1; p.argb = pl->Color.ToARGB();
                p.p[0] = pl->point.x; p.p[1] = pl->point.y; p.p[2] = pl->point.z;
                p.p[3] = pl->normal.x; p.p[4] = pl->normal.y; p.p[5] = pl->normal.z;
            } else if (auto b = std::get_if<AxisAlignedBox>(&it)) {
                p.kind = 4; p.argb = 0xffffffffu;          // its six planes are Color.White (Plane.cs:28)
                p.p[0] = b->Min.x; p.p[1] = b->Min.y; p.p[2] = b->Min.z; p.p[3] = b->Max.x; p.p[4] = b->Max.y; p.p[5] = b->Max.z;
            } else {
                const Triangle& t = std::get<Triangle>(it);
                p.kind = 2; p.argb = t.color;
                const Vector* v[3] = {&t.v1, &t.v2, &t.v3};
                for (int k = 0; k < 3; ++k) { p.p[3 * k] = v[k]->x; p.p[3 * k + 1] = v[k]->y; p.p[3 * k + 2] = v[k]->z; }
            }
            out.push_back(p);
        }
        return out;
    }
private:
    std::vector<Item> items_;
};

}  // namespace Raytrace

// Model after Load3ds + PostProcessGeometry (Model.cs:522-653,750-831): triangles scaled into the unit cube.
class Model {
public:
    bool LoadingComplete = false, LoadingError = false;   // Model.cs:199
    std::vector<double> v9;                                // [n][3][3]
    std::vector<uint32_t> argb;                            // PackColorAndAlpha(diffuse, 1.0), Renderer.cs:1463
    Vector Min, Max;

    void Load3dsModelFromStream(std::istream& stream) {
        std::vector<uint8_t> data((std::istreambuf_iterator<char>(stream)), std::istreambuf_iterator<char>());
        sr_scene* tmp = nullptr;
        sr_check(sr_create(-1, &tmp));                     // host-only handle: parsing is host work
        int rc = sr_load_3ds(tmp, data.data(), data.size());
        if (rc != SR_OK) { sr_destroy(tmp); LoadingError = true; sr_check(rc); }
        int64_t n = sr_num_triangles(tmp);
        v9.resize((size_t)n * 9); argb.resize((size_t)n);
        double mn[3], mx[3];
        sr_get_triangles(tmp, v9.data(), argb.data(), mn, mx);
        sr_destroy(tmp);
        Min = {mn[0], mn[1], mn[2]}; Max = {mx[0], mx[1], mx[2]};
        LoadingComplete = true; LoadingError = false;
    }
    int TriangleCount() const { return (int)argb.size(); }
};

class Instance {                                           // Instance.cs:21-51
public:
    explicit Instance(std::shared_ptr<Model> model) : Model_(std::move(model)) {
        if (!Model_) throw std::invalid_argument("model != null");
    }
    std::shared_ptr<Model> Model_;
    Vector Position{0.0, 0.0, 1.5};
    double Yaw = 0, Pitch = 0, Roll = 0;
    double FieldOfViewDepth = 0.5;
};

class Renderer {
public:
    enum class Style { Standard, ColorShuffle, Negative, DepthSmooth, DepthBanded, Normals, Count };   // Renderer.cs:23-33
    // ---- public fields, Renderer.cs:35-139 ----
    Style RenderStyle = Style::Standard;
    bool depthBuffer = false, depthBufferHires = false;    // only steer the two depth styles here (:837-863)
    double ambientLight_intensity = 0.1;
    Vector directionalLight_dir, positionalLight_pos;
    double specularLight_shininess = 100.0;
    bool pointLighting = true, specularLighting = true;
    bool rayTrace = false, rayTraceShading = true, rayTraceShadows = false, rayTraceShadowsStatic = false;
    bool rayTraceAmbientOcclusion = false, rayTraceLightField = false, rayTraceSubdivision = true;
    bool rayTracePathTracing = false, rayTraceVoxels = false, rayTraceFocalBlur = true;
    double rayTraceFocalDepth = 1.5, rayTraceFocalBlurStrength = 10.0;
    int rayTraceConcurrency = 4, rayTraceSubPixelRes = 1, rayTraceRandomSeed = 1234567890;
    int rayTraceStartRow = 0, rayTraceEndRow = 0;
    // MI355X additions: which structure the device walks (-1: chosen per model, see Mode())
    int gpuTraceMode = -1;
    // How NumGeometryTests / NumNodeVisits / NumLeafNodeVisits (Renderer.cs:476-504) are answered is an explicit choice of the
    // constructor: they are the literal reference-tree traversal's counters, and the fast path does not walk that tree.
    //   Literal  every model's primary rays walk the reference tree (SR_MODE_REF_TREE): the reference's counters, any model size;
    //   Auto     (default) models of fewer than gpuOwnBvhThreshold triangles -- the sizes the reference itself handles -- as Literal,
    //            larger ones as Off;
    //   Off      every subdivided model on the library's own BVH; reading one of the three counters throws InvalidOperationException
    //            (never a silent zero).
    // NumRaysFired is exact in every mode; shadow rays take the shaft path on the own BVH in all three (same pixels).
    enum class TraversalCounters { Auto, Literal, Off };
    const TraversalCounters gpuTraversalCounters;
    int gpuOwnBvhThreshold = 2000;
    int gpuMaxBounces = 0;          // config-5 extension: mirror bounces (0 = the reference's behaviour)
    double gpuReflectivity = 0.0;
    std::vector<std::shared_ptr<Instance>> Instances;
    Raytrace::GeometryCollection ExtraGeometryToRaytrace;

    explicit Renderer(int device = 0, TraversalCounters traversalCounters = TraversalCounters::Auto)   // Renderer.cs:207-230
        : gpuTraversalCounters(traversalCounters) {
        directionalLight_dir = Vector(-1, -1, 1);
        directionalLight_dir.Normalise();
        positionalLight_pos = Vector(0.0, 0.0, 1.5) - directionalLight_dir * 2;
        fieldOfViewDepth_ = sr_default_fov_depth();
        sr_check(sr_create(device, &scene_));
    }
    ~Renderer() { Dispose(); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    void Dispose() { if (scene_) { sr_destroy(scene_); scene_ = nullptr; } }   // Renderer.cs:236

    uint32_t BackgroundColor() const { return backgroundColor_; }
    void BackgroundColor(uint32_t v) { backgroundColor_ = v & 0x00FFFFFFu; }   // Renderer.cs:308-321
    uint32_t BackgroundColorWithAlpha() const { return backgroundColor_ | 0xFF000000u; }

    std::shared_ptr<Engine3D::Model> Model() const { return modelVolatile_; }
    void Model(std::shared_ptr<Engine3D::Model> m) {       // Renderer.cs:349-364
        modelVolatile_ = std::move(m);
        if (modelVolatile_) { modelVolatile_->LoadingComplete = true; modelVolatile_->LoadingError = false; }
    }

    int AntiAliasResolution() const { return antiAliasResolution_; }
    void AntiAliasResolution(int value) {                  // Renderer.cs:366-413
        if (value <= 0) throw std::invalid_argument("AntiAliasResolution must be greater than zero");
        if (value == antiAliasResolution_) return;
        int w = aaPixels_ ? aaWidth_ : width_, h = aaPixels_ ? aaHeight_ : height_;
        int32_t* pixels = aaPixels_ ? aaPixels_ : pixels_;   // restore the original surface
        antiAliasResolution_ = value;
        if (value == 1) {
            aaPixels_ = nullptr;
            hires_.clear();
        } else {
            aaPixels_ = pixels; aaWidth_ = w; aaHeight_ = h;
            w *= value; h *= value;
            hires_.assign((size_t)w * h, 0);               // larger surface for pre-anti-aliased rendering
            pixels = hires_.data();
        }
        SetSurface(w, h, pixels);
    }
    // caller owns `pixels` (int[width*height]); written in place (Renderer.cs:593-626)
    void SetRenderingSurface(int width, int height, int32_t* pixels) {
        const int n = antiAliasResolution_;
        if (width * n == width_ && height * n == height_) {   // unchanged resolution: swap the buffer only
            if (n > 1) aaPixels_ = pixels; else pixels_ = pixels;
            return;
        }
        if (n > 1) {
            aaPixels_ = pixels; aaWidth_ = width; aaHeight_ = height;
            width *= n; height *= n;
            hires_.assign((size_t)width * height, 0);
            pixels = hires_.data();
        }
        SetSurface(width, height, pixels);
    }
    void Load3dsModelFromStream(std::istream& stream) {    // Renderer.cs:629-635
        modelVolatile_ = std::make_shared<Engine3D::Model>();
        modelVolatile_->Load3dsModelFromStream(stream);
    }
    void PreCalculate() {                                  // Renderer.cs:673-699
        if (!PinModel()) throw InvalidOperationException("PreCalculate: no model is loading");
        if (!rayTrace) return;
        if (sceneModel_ != model_.get()) {
            double mn[3] = {model_->Min.x, model_->Min.y, model_->Min.z}, mx[3] = {model_->Max.x, model_->Max.y, model_->Max.z};
            sr_check(sr_set_triangles(scene_, model_->v9.data(), model_->argb.data(), (int64_t)model_->argb.size(), mn, mx));
            sceneModel_ = model_.get();
            built_ = 0;
        }
        int mode = Mode();
        uint32_t want = mode == SR_MODE_BRUTE ? 0u : (1u << mode);
        if (mode == SR_MODE_REF_TREE && rayTraceShadows && !rayTraceShadowsStatic && !model_->argb.empty())
            want |= 1u << SR_MODE_BVH;                     // a tree frame's shadow rays take the BVH's shaft path
        if (want & ~built_) {
            sr_check(sr_build(scene_, want & ~built_, 0, 0));  // SpatialSubdivision defaults 15 / 25
            built_ |= want;
        }
    }
    void Render() {                                        // Renderer.cs:701-778
        if (!rayTrace) throw std::logic_error("the scan-line rasteriser is out of scope of the MI355X hot path");
        if (!PinModel()) return;                           // silently, :736-739
        if (rayTraceAmbientOcclusion || rayTraceLightField || rayTracePathTracing || rayTraceVoxels)
            throw std::logic_error("AO / light field / path tracing / voxels are out of scope (racy or RNG-order dependent in the reference)");
        for (auto& inst : Instances) {
            inst->FieldOfViewDepth = fieldOfViewDepth_;    // :749
            RaytraceGeometry(*inst);
        }
        PostProcessImage();                                // :765
        AntiAliasImage();                                  // :767
    }
    // Renderer.cs:465-504
    int64_t NumRaysFired() const { return (int64_t)stats_[0]; }
    int64_t NumGeometryTests() const { return Counter(1, "NumGeometryTests"); }
    int64_t NumNodeVisits() const { return Counter(2, "NumNodeVisits"); }
    int64_t NumLeafNodeVisits() const { return Counter(3, "NumLeafNodeVisits"); }
    bool TraversalCountersAvailable() const { return haveCounters_; }

    sr_frame BuildFrame(const Instance& instance) const {  // the host half of RaytraceGeometry, :1510-1528,:1652
        sr_frame f{};
        f.width = width_; f.height = height_;
        f.start_row = rayTraceStartRow; f.end_row = rayTraceEndRow;
        f.sub_pixel_res = rayTraceSubPixelRes;
        f.background_argb = backgroundColor_;
        f.flags = (rayTraceShading ? SR_F_SHADING : 0u) | (rayTraceShadows ? SR_F_SHADOWS : 0u) |
                  (rayTraceShadows && rayTraceShadowsStatic ? SR_F_STATIC_SHADOWS : 0u) |      // Renderer.cs:1625; the cache lives in the scene
                  (rayTraceFocalBlur ? SR_F_FOCAL_BLUR : 0u) |
                  (pointLighting ? SR_F_POINT_LIGHT : 0u) | (specularLighting ? SR_F_SPECULAR : 0u) |
                  SR_F_PRIMARY_STATS_ONLY;                 // Num* count primary rays (Renderer.cs:1916-1923)
        f.random_seed = rayTraceRandomSeed;
        f.trace_mode = Mode();
        f.max_bounces = gpuMaxBounces;
        f.concurrency = rayTraceConcurrency;              // fixes the fill order of the static shadow cache
        f.reflectivity = gpuReflectivity;
        double pos[3] = {instance.Position.x, instance.Position.y, instance.Position.z};
        sr_instance_matrices(pos, instance.Yaw, instance.Pitch, instance.Roll, f.transform, f.inv_transform);   // Instance.cs:134-135
        f.position_z = instance.Position.z;
        f.fov_depth = instance.FieldOfViewDepth;
        f.focal_depth = rayTraceFocalDepth; f.focal_blur_strength = rayTraceFocalBlurStrength;
        f.ambient = ambientLight_intensity; f.shininess = specularLight_shininess;
        f.light_dir_view[0] = directionalLight_dir.x; f.light_dir_view[1] = directionalLight_dir.y; f.light_dir_view[2] = directionalLight_dir.z;
        f.light_pos_view[0] = positionalLight_pos.x; f.light_pos_view[1] = positionalLight_pos.y; f.light_pos_view[2] = positionalLight_pos.z;
        return f;
    }

private:
    void SetSurface(int width, int height, int32_t* pixels) {   // Renderer.cs:617-626
        pixels_ = pixels; width_ = width; height_ = height;
        rayTraceStartRow = 0; rayTraceEndRow = height - 1;
    }
    void PostProcessImage() {                              // Renderer.cs:819-897
        if ((RenderStyle == Style::DepthSmooth || RenderStyle == Style::DepthBanded) && (depthBufferHires || !depthBuffer)) return;
        if (RenderStyle == Style::Normals) {
            if (depthBuffer || depthBufferHires) throw std::logic_error("Style.Normals reads the rasteriser's depth buffer (out of scope)");
            return;
        }
        if (RenderStyle != Style::Standard)
            sr_check(sr_post_process(scene_, pixels_, (int64_t)width_ * height_, (int32_t)RenderStyle, backgroundColor_));
    }
    void AntiAliasImage() {                                // Renderer.cs:937-978
        if (antiAliasResolution_ < 2) return;
        sr_check(sr_anti_alias(scene_, pixels_, aaWidth_, aaHeight_, antiAliasResolution_, aaPixels_));
    }
    int64_t Counter(int i, const char* name) const {
        if (!haveCounters_)
            throw InvalidOperationException(std::string(name) + ": the last frame ran on the library's own BVH, which does not produce the reference "
                                            "tree's traversal counters; construct the Renderer with TraversalCounters::Literal (or raise gpuOwnBvhThreshold)");
        return (int64_t)stats_[i];
    }
    bool Literal() const {                                 // do the model's primary rays walk the reference tree (and produce its counters)?
        if (model_ && model_->argb.empty()) return true;   // an empty model: nothing to build a BVH from
        if (gpuTraversalCounters == TraversalCounters::Literal) return true;
        if (gpuTraversalCounters == TraversalCounters::Off) return false;
        return !model_ || (int64_t)model_->argb.size() < (int64_t)gpuOwnBvhThreshold;
    }
    int Mode() const {
        if (gpuTraceMode >= 0) return gpuTraceMode;
        if (!rayTraceSubdivision) return SR_MODE_BRUTE;
        return Literal() ? SR_MODE_REF_TREE : SR_MODE_BVH;
    }
    bool PinModel() {                                      // Renderer.cs:791-810
        if (!modelVolatile_) return false;
        if (modelVolatile_->LoadingComplete) { modelVolatile_->LoadingComplete = false; model_ = modelVolatile_; return true; }
        return model_ != nullptr;
    }
    void RaytraceGeometry(Instance& instance) {            // Renderer.cs:1501-1687
        PreCalculate();
        std::vector<sr_prim> prims = ExtraGeometryToRaytrace.ToPrims();
        sr_check(sr_set_extra_geometry(scene_, prims.data(), (int32_t)prims.size()));
        rayTraceStartRow = std::min(std::max(0, rayTraceStartRow), height_ - 1);   // :1652-1653
        rayTraceEndRow = std::min(std::max(0, rayTraceEndRow), height_ - 1);
        sr_frame f = BuildFrame(instance);
        if (!pixels_) throw InvalidOperationException("SetRenderingSurface must be called before Render");
        if (f.trace_mode != SR_MODE_BVH) {                 // the literal tree (or brute force): the reference's counters
            sr_check(sr_render(scene_, &f, pixels_, stats_));
            haveCounters_ = true;
        } else {                                           // the own BVH does not produce them: reading one throws
            sr_check(sr_render(scene_, &f, pixels_, nullptr));
            const int rows = std::max(0, rayTraceEndRow - rayTraceStartRow + 1);
            stats_[0] = (uint64_t)rows * (uint64_t)width_ * (uint64_t)(rayTraceSubPixelRes * rayTraceSubPixelRes);   // NumRaysFired, :1916
            stats_[1] = stats_[2] = stats_[3] = 0;
            haveCounters_ = false;
        }
    }

    sr_scene* scene_ = nullptr;
    uint32_t backgroundColor_ = 0;
    double fieldOfViewDepth_ = 0;
    int width_ = 1, height_ = 1;
    int32_t* pixels_ = nullptr;
    int antiAliasResolution_ = 1;                          // Renderer.cs:155-156
    int aaWidth_ = 0, aaHeight_ = 0;
    int32_t* aaPixels_ = nullptr;                          // antiAliasedSurface: the caller's buffer while AA > 1
    std::vector<int32_t> hires_;
    std::shared_ptr<Engine3D::Model> modelVolatile_, model_;
    const Engine3D::Model* sceneModel_ = nullptr;
    uint32_t built_ = 0;
    uint64_t stats_[4] = {0, 0, 0, 0};
    bool haveCounters_ = true;                             // before the first frame the reference reads zeros
};

}  // namespace Engine3D
