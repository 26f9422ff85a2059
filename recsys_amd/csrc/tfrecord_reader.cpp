// Streaming TFRecord -> batch reader of librsx.so (host code, no device code).
// Replaces the whole `input_fn` front end of the reference -- tf.data.TFRecordDataset(filenames).map(parse,
// num_parallel_calls).batch(batch_size) of fm/fm.py:106-112 (deepfm/deepfm.py:60-70, xdeepfm/xdeepfm.py:101-118,
// dcn/dcn.py:106-112, din/din.py:61-80) -- at the rate the GPU consumes batches:
//   * files are mmap'ed and every byte is touched ONCE: a scanner thread walks the record framing (12 header bytes per
//     record), a pool of workers verifies the masked CRC-32C (SSE4.2 crc32 instruction) and parses the Example of a
//     record while it is hot in cache, writing the row straight into its slot of the batch buffer;
//   * batches are assembled in C++: `next` copies one finished batch (54 KB at batch 256) into caller memory (pinned
//     host memory on the training path) -- no per-batch numpy slicing, concatenation or Python-side prefetch thread;
//   * records of consecutive files form ONE stream (TFRecordDataset semantics): batches straddle file boundaries, the
//     final partial batch of an epoch is kept (batch before repeat), `num_epochs` < 0 repeats forever;
//   * data-parallel sharding (MirroredStrategy hands successive batches of the one stream to successive replicas,
//     SURVEY.md 8e "replica r gets records r, r+N, ..."): batch b of an epoch belongs to rank b % world; a rank
//     neither verifies nor parses the other ranks' records.  Only complete rounds of `world` full batches are
//     delivered, so every rank runs the same number of equal-size steps (the collectives need that); the < world*bs
//     records left at the end of an epoch are dropped.  drop_remainder == RSX_SHARD_TAIL (2) lifts that for EVALUATION
//     (no collective inside the loop, the metric counters are summed at the end): every batch goes to rank b % world as
//     soon as it is full, the leftover full batches and the final partial one included, so the ranks together cover every
//     record exactly once -- the same examples a single-replica evaluation of the files sees.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rsx.h"

extern "C" uint32_t rsx_masked_crc32c_h(const uint8_t* p, size_t n);
// row parsers of tfrecord_ingest.cpp (one record -> one row of the outputs)
int rsx_criteo_parse_row(const uint8_t* rec, size_t n, const int32_t* slot_src, const int32_t* slot_rows, const float* bnd,
                         const int32_t* bnd_off, const float* shift, int F, uint64_t null_hash, float* label,
                         float* cont_log, int32_t* ids, bool label_optional);
int rsx_din_parse_row(const uint8_t* rec, size_t n, int P, int64_t* label, int64_t* i_id, int64_t* i_cate, int64_t* hist_i,
                      int64_t* hist_c);
extern "C" uint64_t rsx_fingerprint64_h(const uint8_t* s, size_t n);

namespace {

struct Mapped {          // one mmap'ed shard, unmapped when the last batch that points into it is done
  const uint8_t* p = nullptr;
  size_t n = 0;
  ~Mapped() { if (p && n) munmap(const_cast<uint8_t*>(p), n); }
};

struct Rec { const uint8_t* p; uint32_t len; };     // payload; the 4 crc bytes follow it, the 12 header bytes precede it

struct Job {
  int64_t seq = 0;                                  // delivery order on this rank
  std::vector<Rec> recs;
  std::vector<std::shared_ptr<Mapped>> keep;
};

struct Slot {
  std::vector<uint8_t> buf;
  int rows = 0, status = RSX_OK;
  bool ready = false;
};

struct Schema {
  int kind = 0;                                     // 0 criteo, 1 din
  std::vector<int32_t> slot_src, slot_rows, bnd_off;
  std::vector<float> bnd, shift;
  int F = 0, P = 0;
  uint64_t null_hash = 0;
  size_t row_bytes() const {
    return kind == 0 ? 4 + 13 * 4 + (size_t)F * 4 : 3 * 8 + 2 * (size_t)P * 8;
  }
};

}  // namespace

struct rsx_reader {
  Schema sc;
  std::vector<std::string> paths;
  int bs = 0, epochs = -1, rank = 0, world = 1, drop_remainder = 0, verify = 1, nq = 0;
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv_job, cv_slot, cv_free;
  std::deque<Job> jobs;
  int64_t next_deliver = 0;        // seq the consumer takes next
  int64_t produced = 0;            // jobs created so far
  int64_t end_seq = -1;            // total number of jobs once the scanner is done
  int scan_status = RSX_OK;
  bool stop = false;
  std::thread scanner;
  std::vector<std::thread> workers;
  std::atomic<int64_t> records_parsed{0};

  // ---- worker: one whole batch per job --------------------------------------------------------------------------
  void parse_job(Job& j, Slot& s) {
    const size_t rb = sc.row_bytes();
    const int n = (int)j.recs.size();
    s.rows = n;
    s.status = RSX_OK;
    uint8_t* b = s.buf.data();
    // slot layout = the caller's arrays back to back, each sized for `bs` rows
    if (sc.kind == 0) {
      float* label = reinterpret_cast<float*>(b);
      float* cont = reinterpret_cast<float*>(b + (size_t)bs * 4);
      int32_t* ids = reinterpret_cast<int32_t*>(b + (size_t)bs * 4 * 14);
      for (int r = 0; r < n; ++r) {
        const Rec& rc = j.recs[r];
        if (verify) {
          uint32_t c;
          std::memcpy(&c, rc.p + rc.len, 4);
          if (c != rsx_masked_crc32c_h(rc.p, rc.len)) { s.status = RSX_EDATA; continue; }
        }
        const int st = rsx_criteo_parse_row(rc.p, rc.len, sc.slot_src.data(), sc.slot_rows.data(), sc.bnd.data(),
                                            sc.bnd_off.data(), sc.shift.data(), sc.F, sc.null_hash, label + r,
                                            cont + (size_t)r * 13, ids + (size_t)r * sc.F, false);
        if (st != RSX_OK) s.status = st;
      }
    } else {
      int64_t* label = reinterpret_cast<int64_t*>(b);
      int64_t* iid = label + bs;
      int64_t* icat = iid + bs;
      int64_t* hi = icat + bs;
      int64_t* hc = hi + (size_t)bs * sc.P;
      for (int r = 0; r < n; ++r) {
        const Rec& rc = j.recs[r];
        if (verify) {
          uint32_t c;
          std::memcpy(&c, rc.p + rc.len, 4);
          if (c != rsx_masked_crc32c_h(rc.p, rc.len)) { s.status = RSX_EDATA; continue; }
        }
        const int st = rsx_din_parse_row(rc.p, rc.len, sc.P, label + r, iid + r, icat + r, hi + (size_t)r * sc.P,
                                         hc + (size_t)r * sc.P);
        if (st != RSX_OK) s.status = st;
      }
    }
    (void)rb;
    records_parsed.fetch_add(n, std::memory_order_relaxed);
  }

  void worker_loop() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
        if (stop) return;
        j = std::move(jobs.front());
        jobs.pop_front();
      }
      Slot& s = slots[j.seq % nq];          // the scanner only creates job seq when slot seq % nq is free
      parse_job(j, s);
      j.keep.clear();
      {
        std::lock_guard<std::mutex> lk(mu);
        s.ready = true;
      }
      cv_slot.notify_all();
    }
  }

  // ---- scanner: framing walk, batch formation, sharding ---------------------------------------------------------
  bool push_job(Job&& j) {                   // blocks while the ring is full; false when stopping
    std::unique_lock<std::mutex> lk(mu);
    cv_free.wait(lk, [&] { return stop || produced - next_deliver < nq; });
    if (stop) return false;
    j.seq = produced++;
    jobs.push_back(std::move(j));
    lk.unlock();
    cv_job.notify_one();
    return true;
  }

  void finish_scan(int status) {
    {
      std::lock_guard<std::mutex> lk(mu);
      scan_status = status;
      end_seq = produced;
    }
    cv_slot.notify_all();
  }

  void scanner_loop() {
    for (int ep = 0; epochs < 0 || ep < epochs; ++ep) {
      int64_t batch_in_epoch = 0;            // index of the batch being filled, over ALL ranks
      int in_batch = 0;                      // records of that batch seen so far
      Job cur;                               // my batch of the current round while it fills
      Job held;                              // my complete batch, waiting for its round to complete (world > 1)
      bool have_held = false;
      int64_t total_records = 0;
      for (const std::string& path : paths) {
        auto m = std::make_shared<Mapped>();
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) { finish_scan(RSX_EDATA); return; }
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); finish_scan(RSX_EDATA); return; }
        m->n = (size_t)st.st_size;
        if (m->n) {
          void* a = mmap(nullptr, m->n, PROT_READ, MAP_PRIVATE, fd, 0);
          if (a == MAP_FAILED) { close(fd); m->n = 0; finish_scan(RSX_EDATA); return; }
          m->p = static_cast<const uint8_t*>(a);
          madvise(a, m->n, MADV_SEQUENTIAL);
        }
        close(fd);
        size_t p = 0;
        const size_t n = m->n;
        while (p < n) {
          if (n - p < 12) { finish_scan(RSX_EDATA); return; }
          uint64_t len;
          std::memcpy(&len, m->p + p, 8);
          if (verify) {
            uint32_t c;
            std::memcpy(&c, m->p + p + 8, 4);
            if (c != rsx_masked_crc32c_h(m->p + p, 8)) { finish_scan(RSX_EDATA); return; }
          }
          if (len > n - p - 12 || n - p - 12 - len < 4 || len > 0xFFFFFFFFull) { finish_scan(RSX_EDATA); return; }
          // This walk is the one serial stage of the reader and it is a chain of cache misses (one header per record, the
          // next address known only from this length): records of one dataset have nearly the same size, so the lines where
          // the headers 4 and 5 records ahead will most likely sit are requested now.
          {
            const size_t step = 16 + (size_t)len, a4 = p + 4 * step, a5 = a4 + step;
            if (a5 + 64 < n) {
              __builtin_prefetch(m->p + a4);
              __builtin_prefetch(m->p + a5);
            }
          }
          const bool mine = (batch_in_epoch % world) == rank;
          if (mine) {
            if (cur.keep.empty() || cur.keep.back().get() != m.get()) cur.keep.push_back(m);
            if (cur.recs.empty()) cur.recs.reserve((size_t)bs);
            cur.recs.push_back({m->p + p + 12, (uint32_t)len});
          }
          p += 16 + len;
          ++total_records;
          if (++in_batch == bs) {            // batch `batch_in_epoch` is complete
            if (mine) {
              if (world == 1 || drop_remainder == RSX_SHARD_TAIL) { if (!push_job(std::move(cur))) return; }
              else { held = std::move(cur); have_held = true; }
              cur = Job();
            }
            if (world > 1 && batch_in_epoch % world == world - 1 && have_held) {     // the round is complete
              if (!push_job(std::move(held))) return;
              held = Job();
              have_held = false;
            }
            in_batch = 0;
            ++batch_in_epoch;
          }
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          if (stop) return;
        }
      }
      // end of the epoch: the final partial batch is kept on a single replica (batch before repeat, fm/fm.py:108-111);
      // data-parallel runs deliver complete rounds only (see the header).
      if (in_batch > 0 && ((world == 1 && !drop_remainder) ||
                           (drop_remainder == RSX_SHARD_TAIL && (batch_in_epoch % world) == rank))) {
        if (!push_job(std::move(cur))) return;
      }
      if (total_records == 0) break;         // nothing to repeat
    }
    finish_scan(RSX_OK);
  }

  // ---- consumer ----------------------------------------------------------------------------------------------
  int next(uint8_t* const* outs, const size_t* row_bytes, int nout) {
    std::unique_lock<std::mutex> lk(mu);
    const int64_t seq = next_deliver;
    Slot& s = slots[seq % nq];
    cv_slot.wait(lk, [&] { return s.ready || (end_seq >= 0 && seq >= end_seq); });
    if (!s.ready) return scan_status != RSX_OK ? scan_status : 0;      // end of data (or the scanner failed)
    lk.unlock();
    const int rows = s.rows, status = s.status;
    size_t off = 0;
    for (int i = 0; i < nout; ++i) {
      if (outs[i]) std::memcpy(outs[i], s.buf.data() + off, row_bytes[i] * (size_t)rows);
      off += row_bytes[i] * (size_t)bs;
    }
    lk.lock();
    s.ready = false;
    ++next_deliver;
    lk.unlock();
    cv_free.notify_all();
    return status != RSX_OK ? status : rows;
  }

  void start(int threads, int queue_batches) {
    nq = queue_batches < 2 ? 2 : queue_batches;
    slots.resize(nq);
    for (auto& s : slots) s.buf.resize(sc.row_bytes() * (size_t)bs + 64);
    scanner = std::thread([this] { scanner_loop(); });
    if (threads < 1) threads = 1;
    for (int t = 0; t < threads; ++t) workers.emplace_back([this] { worker_loop(); });
  }

  ~rsx_reader() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_job.notify_all();
    cv_free.notify_all();
    cv_slot.notify_all();
    if (scanner.joinable()) scanner.join();
    for (auto& w : workers) if (w.joinable()) w.join();
  }
};

static rsx_reader* open_common(const char* const* paths_h, int n_paths, int batch_size, int num_epochs, int shard_rank,
                               int shard_world, int drop_remainder, int verify_crc) {
  if (!paths_h || n_paths <= 0 || batch_size <= 0 || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world)
    return nullptr;
  auto* r = new rsx_reader();
  for (int i = 0; i < n_paths; ++i) {
    if (!paths_h[i]) { delete r; return nullptr; }
    r->paths.emplace_back(paths_h[i]);
  }
  r->bs = batch_size;
  r->epochs = num_epochs;
  r->rank = shard_rank;
  r->world = shard_world;
  r->drop_remainder = drop_remainder;
  r->verify = verify_crc;
  return r;
}

extern "C" rsx_reader* rsx_criteo_reader_open_h(const char* const* paths_h, int n_paths, const int32_t* slot_src_h,
                                                const int32_t* slot_rows_h, const float* bnd_h, const int32_t* bnd_off_h,
                                                const float* shift_h, int F, int batch_size, int num_epochs,
                                                int shard_rank, int shard_world, int drop_remainder, int threads,
                                                int verify_crc, int queue_batches) {
  if (!slot_src_h || !slot_rows_h || !bnd_off_h || !shift_h || F <= 0 || F > 64) return nullptr;
  rsx_reader* r = open_common(paths_h, n_paths, batch_size, num_epochs, shard_rank, shard_world, drop_remainder, verify_crc);
  if (!r) return nullptr;
  r->sc.kind = 0;
  r->sc.F = F;
  r->sc.slot_src.assign(slot_src_h, slot_src_h + F);
  r->sc.slot_rows.assign(slot_rows_h, slot_rows_h + F);
  r->sc.bnd_off.assign(bnd_off_h, bnd_off_h + F + 1);
  const int nb = bnd_off_h[F];
  r->sc.bnd.assign(nb > 0 ? bnd_h : nullptr, nb > 0 ? bnd_h + nb : nullptr);
  if (r->sc.bnd.empty()) r->sc.bnd.push_back(0.f);
  r->sc.shift.assign(shift_h, shift_h + 13);
  r->sc.null_hash = rsx_fingerprint64_h(reinterpret_cast<const uint8_t*>("NULL"), 4);
  r->start(threads, queue_batches);
  return r;
}

extern "C" rsx_reader* rsx_din_reader_open_h(const char* const* paths_h, int n_paths, int P, int batch_size, int num_epochs,
                                             int shard_rank, int shard_world, int drop_remainder, int threads,
                                             int verify_crc, int queue_batches) {
  if (P <= 0) return nullptr;
  rsx_reader* r = open_common(paths_h, n_paths, batch_size, num_epochs, shard_rank, shard_world, drop_remainder, verify_crc);
  if (!r) return nullptr;
  r->sc.kind = 1;
  r->sc.P = P;
  r->start(threads, queue_batches);
  return r;
}

extern "C" int rsx_criteo_reader_next_h(rsx_reader* r, float* label_h, float* cont_log_h, int32_t* ids_h) {
  if (!r || r->sc.kind != 0 || !label_h || !ids_h) return RSX_EINVAL;
  uint8_t* outs[3] = {reinterpret_cast<uint8_t*>(label_h), reinterpret_cast<uint8_t*>(cont_log_h),
                      reinterpret_cast<uint8_t*>(ids_h)};
  const size_t rb[3] = {4, 13 * 4, (size_t)r->sc.F * 4};
  return r->next(outs, rb, 3);
}

extern "C" int rsx_din_reader_next_h(rsx_reader* r, int64_t* label_h, int64_t* i_id_h, int64_t* i_cate_h, int64_t* hist_i_h,
                                     int64_t* hist_c_h) {
  if (!r || r->sc.kind != 1 || !label_h || !i_id_h || !i_cate_h || !hist_i_h || !hist_c_h) return RSX_EINVAL;
  uint8_t* outs[5] = {reinterpret_cast<uint8_t*>(label_h), reinterpret_cast<uint8_t*>(i_id_h),
                      reinterpret_cast<uint8_t*>(i_cate_h), reinterpret_cast<uint8_t*>(hist_i_h),
                      reinterpret_cast<uint8_t*>(hist_c_h)};
  const size_t rb[5] = {8, 8, 8, (size_t)r->sc.P * 8, (size_t)r->sc.P * 8};
  return r->next(outs, rb, 5);
}

extern "C" int64_t rsx_reader_records_parsed_h(const rsx_reader* r) { return r ? r->records_parsed.load() : 0; }

extern "C" void rsx_reader_close_h(rsx_reader* r) { delete r; }
