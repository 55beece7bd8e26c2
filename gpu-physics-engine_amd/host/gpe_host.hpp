// gpe_host.hpp -- C++ host side above the C-ABI (include/gpe.h), mirroring the reference's Rust module API.
//
// The reference's host is Rust (no toolchain in this image), so the compiled-language mirror is C++: same type and
// method names, same argument meaning and the same error behaviour (the reference unwrap()s / panics -- here every
// failing gpe_status throws gpe::Error with gpe_last_error()).  tests/cpp/reference_tests.cpp restates the reference's
// five integration test files on top of it.  Citations: /root/reference/src.
//
//   WgpuContext::new_for_test  -> gpe::Context          renderer/wgpu_context.rs:73-101 (device + queue == HIP stream)
//   GpuBuffer<T>               -> gpe::GpuBuffer<T>     utils/gpu_buffer.rs:7-29   (device buffer + host mirror)
//   ParticleSystem             -> gpe::ParticleSystem   particles/particle_system.rs:16-24
//   Grid                       -> gpe::Grid             grid/grid.rs:24-33
//   CollisionSystem            -> gpe::CollisionSystem  physics/collision_system.rs:9-12
//   GPUSorter / PushConstants  -> gpe::GPUSorter        utils/radix_sort/radix_sort.rs:44-58
//   PrefixSum                  -> gpe::PrefixSum        utils/prefix_sum/prefix_sum.rs:11-18
//   State                      -> gpe::State            state.rs:21-31 (update() == gpe_step)
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gpe.h"

namespace gpe {

struct Vec2 { float x, y; };                       // glam::Vec2
inline bool operator==(const Vec2 &a, const Vec2 &b) { return a.x == b.x && a.y == b.y; }

constexpr uint32_t UNUSED_CELL_ID = GPE_UNUSED_CELL_ID;            // grid.rs:22
constexpr uint32_t MAX_CELLS_PER_OBJECT = GPE_MAX_CELLS_PER_OBJECT;  // grid.rs:18
constexpr uint32_t NUM_BLOCKS_PER_WORKGROUP = 45;                   // radix_sort.rs:40
constexpr uint32_t RADIX_SORT_BUCKETS = 256;                        // radix_sort.rs:30
constexpr uint32_t WORKGROUP_SIZE = 256;                            // radix_sort.rs:21

class Error : public std::runtime_error {
   public:
    Error(gpe_status s, const std::string &m) : std::runtime_error("gpe status " + std::to_string(s) + ": " + m), status(s) {}
    gpe_status status;
};

class Context {
   public:
    explicit Context(Vec2 world = {1920.0f, 1080.0f}, uint32_t mode = GPE_MODE_COMPAT)
    {
        gpe_config cfg;
        check(gpe_config_default(&cfg), nullptr);
        cfg.world_width = world.x;
        cfg.world_height = world.y;
        cfg.mode = mode;
        check(gpe_create(&cfg, &ctx_), nullptr);
    }
    ~Context() { if (ctx_) gpe_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    gpe_ctx *raw() const { return ctx_; }
    void call(gpe_status s) const { check(s, ctx_); }
    static void check(gpe_status s, const gpe_ctx *c)
    {
        if (s != GPE_OK) throw Error(s, gpe_last_error(c));
    }
    template <typename T>
    std::vector<T> download(gpe_array what) const
    {
        uint64_t bytes = 0;
        call(gpe_array_bytes(ctx_, what, &bytes));
        std::vector<T> out(bytes / sizeof(T));
        call(gpe_download(ctx_, what, out.data(), bytes));
        return out;
    }

   private:
    gpe_ctx *ctx_ = nullptr;
};

// utils/gpu_buffer.rs:7-29 : a device buffer plus its host mirror (`data()`); download() refreshes the mirror.  Like the
// reference's, the device buffer grows by doubling (gpu_buffer.rs:54-76) and keeps its DEVICE contents when it does.
template <typename T>
class GpuBuffer {
   public:
    GpuBuffer(const Context &ctx, std::vector<T> data) : ctx_(&ctx), data_(std::move(data))
    {
        cap_bytes_ = std::max<size_t>(1, data_.size() * sizeof(T));
        ctx_->call(gpe_buffer_alloc(ctx_->raw(), cap_bytes_, &dptr_));
        ctx_->call(gpe_buffer_upload(ctx_->raw(), dptr_, data_.data(), data_.size() * sizeof(T)));
    }
    ~GpuBuffer() { if (dptr_) gpe_buffer_free(ctx_->raw(), dptr_); }
    GpuBuffer(const GpuBuffer &) = delete;
    GpuBuffer &operator=(const GpuBuffer &) = delete;
    GpuBuffer(GpuBuffer &&o) noexcept : ctx_(o.ctx_), data_(std::move(o.data_)), dptr_(o.dptr_), cap_bytes_(o.cap_bytes_) { o.dptr_ = nullptr; }
    size_t len() const { return data_.size(); }
    const std::vector<T> &data() const { return data_; }
    size_t capacity_bytes() const { return cap_bytes_; }
    T *device() const { return static_cast<T *>(dptr_); }
    const std::vector<T> &download()                                   // gpu_buffer.rs:96-175
    {
        ctx_->call(gpe_buffer_download(ctx_->raw(), dptr_, data_.data(), data_.size() * sizeof(T)));
        return data_;
    }
    // gpu_buffer.rs:177-262: the last element as the DEVICE holds it; false for an empty buffer.  The mirror stays.
    bool download_last(T *out) const
    {
        if (data_.empty()) return false;
        ctx_->call(gpe_buffer_download(ctx_->raw(), device() + (data_.size() - 1), out, sizeof(T)));
        return true;
    }
    void push(const T &value) { append(&value, 1); }                   // gpu_buffer.rs:30-33
    void push_all(const std::vector<T> &values) { append(values.data(), values.size()); }   // gpu_buffer.rs:35-38
    void replace_elem(const T &new_data, size_t index)                 // gpu_buffer.rs:264-275 (the reference panics)
    {
        if (index >= data_.size()) throw std::out_of_range("Index out of bounds");
        data_[index] = new_data;
        ctx_->call(gpe_buffer_upload(ctx_->raw(), device() + index, &new_data, sizeof(T)));
    }

   private:
    // gpu_buffer.rs:49-87 (`upload`): a buffer that is too small is replaced by one of twice the needed size, the old
    // device contents are carried over (buffer-to-buffer copy there; through the host here: the C-ABI has no
    // device-to-device copy and this is off the step path), then only the new tail is written
    void append(const T *values, size_t count)
    {
        const size_t old_n = data_.size(), need = (old_n + count) * sizeof(T);
        if (need > cap_bytes_) {
            std::vector<T> kept(old_n);
            ctx_->call(gpe_buffer_download(ctx_->raw(), dptr_, kept.data(), old_n * sizeof(T)));
            void *fresh = nullptr;
            const size_t cap = 2 * std::max<size_t>(need, 1);
            ctx_->call(gpe_buffer_alloc(ctx_->raw(), cap, &fresh));
            ctx_->call(gpe_buffer_upload(ctx_->raw(), fresh, kept.data(), old_n * sizeof(T)));
            ctx_->call(gpe_buffer_free(ctx_->raw(), dptr_));
            dptr_ = fresh;
            cap_bytes_ = cap;
        }
        data_.insert(data_.end(), values, values + count);
        ctx_->call(gpe_buffer_upload(ctx_->raw(), device() + old_n, values, count * sizeof(T)));
    }
    const Context *ctx_;
    std::vector<T> data_;
    void *dptr_ = nullptr;
    size_t cap_bytes_ = 0;
};

// particles/particle_buffers.rs:4-10 (host copies as downloaded)
struct ParticleBuffers {
    std::vector<Vec2> current_positions, previous_positions;
    std::vector<float> radii;
    std::vector<uint32_t> home_cell_ids;
};

class ParticleSystem {
   public:
    // particle_system.rs:49-99 (previous = current; max_radius = the radius of largest magnitude)
    static ParticleSystem new_from_buffers(const Context &ctx, const std::vector<Vec2> &positions,
                                           const std::vector<float> &radii)
    {
        if (positions.size() != radii.size()) throw std::invalid_argument("positions and radii differ in length");
        ctx.call(gpe_set_particles(ctx.raw(), &positions[0].x, nullptr, radii.data(), positions.size()));
        return ParticleSystem(ctx);
    }
    void add_particles(const std::vector<Vec2> &positions, const std::vector<float> &radii)   // :163-220
    {
        ctx_->call(gpe_add_particles(ctx_->raw(), &positions[0].x, radii.data(), positions.size()));
    }
    size_t len() const { uint64_t n = 0; ctx_->call(gpe_len(ctx_->raw(), &n)); return n; }               // :275
    float get_max_radius() const { float r = 0; ctx_->call(gpe_max_radius(ctx_->raw(), &r)); return r; } // :291
    void sort_by_cell_id(float /*cell_size: the Grid's, state.rs:123*/) { ctx_->call(gpe_morton_resort(ctx_->raw())); }
    void update_positions(float dt) { ctx_->call(gpe_integrate(ctx_->raw(), dt)); }                      // :245
    void mouse_click_callback(bool pressed, Vec2 p)                                                      // :221-224
    {
        mouse_pressed_ = pressed;
        ctx_->call(gpe_set_mouse(ctx_->raw(), pressed, p.x, p.y));
    }
    void mouse_move_callback(Vec2 p) { ctx_->call(gpe_set_mouse(ctx_->raw(), mouse_pressed_, p.x, p.y)); } // :225-227
    // the reference's wall-clock re-sort policy (SORT_INTERVAL = 4 s, :13-14; the first frame sorts, :45,97)
    bool is_it_time_to_sort() const                                                                      // :229-231
    {
        return !sorted_once_ || std::chrono::steady_clock::now() - last_sort_time_ >= std::chrono::seconds(4);
    }
    void reset_last_sort_time() { last_sort_time_ = std::chrono::steady_clock::now(); sorted_once_ = true; } // :233-235
    std::vector<uint32_t> download_home_cell_ids() const { return ctx_->download<uint32_t>(GPE_HOME_CELL_IDS); }
    std::vector<uint32_t> download_particle_ids() const { return ctx_->download<uint32_t>(GPE_PARTICLE_IDS); }
    ParticleBuffers download_particle_buffers() const                                                    // :258-265
    {
        ParticleBuffers b;
        b.current_positions = ctx_->download<Vec2>(GPE_POS);
        b.previous_positions = ctx_->download<Vec2>(GPE_PREV);
        b.radii = ctx_->download<float>(GPE_RADIUS);
        b.home_cell_ids = ctx_->download<uint32_t>(GPE_HOME_CELL_IDS);
        return b;
    }

   private:
    explicit ParticleSystem(const Context &ctx) : ctx_(&ctx) {}
    const Context *ctx_;
    bool mouse_pressed_ = false;
    bool sorted_once_ = false;
    std::chrono::steady_clock::time_point last_sort_time_{};
};

class Grid {
   public:
    // grid.rs:74-149 : cell size from an explicit max radius (tests) ...
    static Grid new_without_camera(const Context &ctx, float max_obj_radius, const ParticleSystem &)
    {
        ctx.call(gpe_grid_set_max_radius(ctx.raw(), max_obj_radius));
        return Grid(ctx);
    }
    // ... or from the particle system's (grid.rs:66-71)
    Grid(const Context &ctx, const ParticleSystem &) : ctx_(&ctx) {}
    static float compute_cell_size(float max_obj_radius) { return gpe_compute_cell_size(max_obj_radius); }   // :159
    float cell_size() const { float cs = 0; ctx_->call(gpe_cell_size(ctx_->raw(), &cs)); return cs; }
    void build_cell_ids() { ctx_->call(gpe_grid_build(ctx_->raw())); }        // :296-306
    void sort_map() { ctx_->call(gpe_grid_sort(ctx_->raw())); }               // :310-312
    void update() { ctx_->call(gpe_grid_update(ctx_->raw())); }               // :322-332
    std::vector<uint32_t> download_cell_ids() const { return ctx_->download<uint32_t>(GPE_CELL_IDS); }       // :314
    std::vector<uint32_t> download_object_ids() const { return ctx_->download<uint32_t>(GPE_OBJECT_IDS); }   // :318

   private:
    explicit Grid(const Context &ctx) : ctx_(&ctx) {}
    const Context *ctx_;
};

class CollisionSystem {
   public:
    CollisionSystem(const Context &ctx, uint32_t dim, const ParticleSystem &, const Grid &) : ctx_(&ctx)   // :14-22
    {
        if (dim != 2) throw std::invalid_argument("2-D only (state.rs:18)");
    }
    void solve_collisions() { ctx_->call(gpe_solve_collisions(ctx_->raw())); }                             // :30-39
    std::vector<uint32_t> download_collision_cells() const { return ctx_->download<uint32_t>(GPE_COLLISION_CELLS); }

   private:
    const Context *ctx_;
};

struct PushConstants { uint32_t num_elements, current_shift, num_workgroups, num_blocks_per_workgroup; };   // radix_sort.rs:53-58

class GPUSorter {
   public:
    GPUSorter(const Context &ctx, uint32_t length, GpuBuffer<uint32_t> &keys, GpuBuffer<uint32_t> &payload)  // :61
        : ctx_(&ctx), length_(length), keys_(&keys), payload_(&payload), keys_b_(ctx, std::vector<uint32_t>(length, 0)),
          payload_b_(ctx, std::vector<uint32_t>(length, 0)), histogram_(ctx, std::vector<uint32_t>(RADIX_SORT_BUCKETS, 0))
    {
        if (length == 0) throw std::invalid_argument("NonZeroU32 length");
    }
    void sort(const uint32_t *sort_first_n = nullptr)                                                        // :199-217
    {
        ctx_->call(gpe_sort_pairs_u32(ctx_->raw(), keys_->device(), payload_->device(), sort_first_n ? *sort_first_n : length_));
    }
    void build_histogram(const PushConstants &pc, bool /*ping*/)                                             // :180-188
    {
        ctx_->call(gpe_sort_histogram_u32(ctx_->raw(), keys_->device(), pc.num_elements, pc.current_shift, histogram_.device()));
    }
    void scatter(const PushConstants &pc, bool /*ping*/)                                                     // :190-198
    {
        ctx_->call(gpe_sort_scatter_pass_u32(ctx_->raw(), keys_->device(), payload_->device(), keys_b_.device(),
                                             payload_b_.device(), pc.num_elements, pc.current_shift));
    }
    const std::vector<uint32_t> &get_keys_b() { return keys_b_.download(); }                                 // :219
    const std::vector<uint32_t> &get_histogram() { return histogram_.download(); }                           // :223

   private:
    const Context *ctx_;
    uint32_t length_;
    GpuBuffer<uint32_t> *keys_, *payload_;
    GpuBuffer<uint32_t> keys_b_, payload_b_, histogram_;
};

class PrefixSum {
   public:
    PrefixSum(const Context &ctx, GpuBuffer<uint32_t> &buffer) : ctx_(&ctx), buffer_(&buffer) {}             // :21
    void execute(uint32_t num_items) { ctx_->call(gpe_inclusive_scan_u32(ctx_->raw(), buffer_->device(), num_items)); }  // :143-160
    void update_buffers(GpuBuffer<uint32_t> &buffer) { buffer_ = &buffer; }                                 // :172

   private:
    const Context *ctx_;
    GpuBuffer<uint32_t> *buffer_;
};

// state.rs:21-31 without window / renderer
class State {
   public:
    State(const std::vector<Vec2> &positions, const std::vector<float> &radii, Vec2 world, uint32_t mode = GPE_MODE_NATIVE)
        : ctx_(world, mode), particles_(ParticleSystem::new_from_buffers(ctx_, positions, radii)), grid_(ctx_, particles_),
          collision_system_(ctx_, 2, particles_, grid_) {}
    void update(float dt, bool resort) { ctx_.call(gpe_step(ctx_.raw(), dt, resort ? GPE_STEP_RESORT : 0u)); }   // state.rs:115-131
    // ... with the reference's own wall-clock re-sort policy (state.rs:122-125, particle_system.rs:229-235)
    bool update(float dt)
    {
        const bool resort = particles_.is_it_time_to_sort();
        update(dt, resort);
        if (resort) particles_.reset_last_sort_time();
        return resort;
    }
    gpe_pipeline_info pipeline_info() const
    {
        gpe_pipeline_info info{};
        info.struct_size = sizeof(info);
        ctx_.call(gpe_get_pipeline_info(ctx_.raw(), &info));
        return info;
    }
    ParticleSystem &particles() { return particles_; }
    Grid &grid() { return grid_; }
    CollisionSystem &collision_system() { return collision_system_; }
    const Context &context() const { return ctx_; }

   private:
    Context ctx_;
    ParticleSystem particles_;
    Grid grid_;
    CollisionSystem collision_system_;
};

}  // namespace gpe
