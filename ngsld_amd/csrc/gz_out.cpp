// gz_out.cpp -- gzip-compressed output at the rate the engine produces text ("optional .gz output"; the
// reference writes plain text only, ngsLD.cpp:73-75).  ngsld_host_gz_open hands out the write end of a pipe: whatever is
// written to it -- the header, device-formatted batches, the host formatter's rows -- is cut into blocks of 4 MiB, every
// block is deflated on its own by one of n_threads workers into a complete gzip member, and the members are written to
// the file in order.  A concatenation of gzip members is a valid .gz file (gzip -d, zcat, zlib's gzread all read through
// it), so the output is what `ngsLD ... | gzip` would produce in content, at n_threads times the speed.
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/ngsld_host.h"

struct ngsld_gz {
  int fd_out = -1, fd_read = -1;  // (the pipe's write end belongs to the caller from ngsld_host_gz_open on)
  int level = 1;
  size_t block = 4u << 20;
  struct Slot {
    std::vector<unsigned char> in, out;
    size_t n_in = 0, n_out = 0;
    uint64_t index = 0;
    int state = 0;  // 0 free, 1 filled, 2 deflated
  };
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t next_fill = 0, next_take = 0, next_write = 0;
  bool eof = false, failed = false;
  std::vector<std::thread> workers;
  std::thread reader, writer;
};

namespace {

bool deflate_member(ngsld_gz::Slot &s, int level) {
  z_stream z;
  std::memset(&z, 0, sizeof(z));
  if (deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;  // + 16: gzip wrapper
  const size_t bound = deflateBound(&z, (uLong)s.n_in) + 64;
  if (s.out.size() < bound) s.out.resize(bound);
  z.next_in = s.in.data();
  z.avail_in = (uInt)s.n_in;
  z.next_out = s.out.data();
  z.avail_out = (uInt)s.out.size();
  const int rc = deflate(&z, Z_FINISH);
  s.n_out = s.out.size() - z.avail_out;
  deflateEnd(&z);
  return rc == Z_STREAM_END;
}

// After a failure (a write error such as ENOSPC / EIO, a deflate error) the pipe is still read -- and its content thrown
// away -- until the producer closes its end: a producer blocked in write() on a full pipe that nobody drains would never
// get to close it, and the process would hang holding the device.  ngsld_host_gz_close then reports the failure.
void drain_pipe(ngsld_gz *g) {
  std::vector<unsigned char> scratch(1u << 20);
  for (;;) {
    const ssize_t r = ::read(g->fd_read, scratch.data(), scratch.size());
    if (r == 0) break;
    if (r < 0 && errno != EINTR) {
      // the pipe cannot be read any more: close the read end, so that a producer blocked in write() on the full pipe gets
      // EPIPE (SIGPIPE is the caller's to ignore or handle) instead of waiting for a reader that has gone
      ::close(g->fd_read);
      g->fd_read = -1;
      break;
    }
  }
}

void reader_loop(ngsld_gz *g) {
  const size_t n_slots = g->slots.size();
  bool failed = false;
  for (;;) {
    ngsld_gz::Slot *s;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      s = &g->slots[g->next_fill % n_slots];
      g->cv.wait(lk, [&] { return s->state == 0 || g->failed; });
      if (g->failed) {
        failed = true;
        break;
      }
    }
    size_t got = 0;
    while (got < g->block) {
      const ssize_t r = ::read(g->fd_read, s->in.data() + got, g->block - got);
      if (r < 0 && errno == EINTR) continue;
      if (r < 0) {
        std::lock_guard<std::mutex> lk(g->mu);
        g->failed = true;
        break;
      }
      if (r == 0) break;
      got += (size_t)r;
    }
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->failed) {
      failed = true;
      break;
    }
    if (got == 0) break;
    s->n_in = got;
    s->index = g->next_fill++;
    s->state = 1;
    g->cv.notify_all();
    if (got < g->block) {  // the write end was closed
      g->eof = true;
      break;
    }
  }
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->eof = true;
    g->cv.notify_all();
  }
  if (failed) drain_pipe(g);
}

void worker_loop(ngsld_gz *g) {
  const size_t n_slots = g->slots.size();
  for (;;) {
    ngsld_gz::Slot *s;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv.wait(lk, [&] { return g->failed || g->next_take < g->next_fill || (g->eof && g->next_take >= g->next_fill); });
      if (g->failed || g->next_take >= g->next_fill) return;
      s = &g->slots[g->next_take++ % n_slots];
    }
    const bool ok = deflate_member(*s, g->level);
    std::lock_guard<std::mutex> lk(g->mu);
    if (!ok) g->failed = true;
    s->state = 2;
    g->cv.notify_all();
  }
}

void writer_loop(ngsld_gz *g) {
  const size_t n_slots = g->slots.size();
  for (;;) {
    ngsld_gz::Slot *s;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      s = &g->slots[g->next_write % n_slots];
      g->cv.wait(lk, [&] { return g->failed || (s->state == 2 && s->index == g->next_write) || (g->eof && g->next_write >= g->next_fill); });
      if (g->failed) return;
      if (!(s->state == 2 && s->index == g->next_write)) return;  // everything written
    }
    const unsigned char *q = s->out.data();
    size_t left = s->n_out;
    bool bad = false;
    while (left && !bad) {
      const ssize_t w = ::write(g->fd_out, q, left);
      if (w < 0 && errno == EINTR) continue;
      if (w <= 0) bad = true;
      else { q += w; left -= (size_t)w; }
    }
    std::lock_guard<std::mutex> lk(g->mu);
    if (bad) g->failed = true;
    s->state = 0;
    ++g->next_write;
    g->cv.notify_all();
  }
}

}  // namespace

extern "C" {

int ngsld_host_gz_open(const char *path, int n_threads, ngsld_gz **out, int *fd_to_write) try {
  if (path == nullptr || out == nullptr || fd_to_write == nullptr) return NGSLD_ERR_INVALID;
  *out = nullptr;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  ngsld_gz *g = new ngsld_gz();
  if (const char *e = std::getenv("NGSLD_GZ_LEVEL")) {
    const int l = std::atoi(e);
    if (l >= 1 && l <= 9) g->level = l;
  }
  g->fd_out = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  int p[2] = {-1, -1};
  if (g->fd_out < 0 || ::pipe(p) != 0) {
    if (g->fd_out >= 0) ::close(g->fd_out);
    delete g;
    return NGSLD_ERR_INVALID;
  }
#ifdef F_SETPIPE_SZ
  (void)fcntl(p[1], F_SETPIPE_SZ, 1 << 20);
#endif
  g->fd_read = p[0];
  g->slots.resize((size_t)n_threads * 2 + 2);
  for (auto &s : g->slots) s.in.resize(g->block);
  g->reader = std::thread(reader_loop, g);
  g->writer = std::thread(writer_loop, g);
  for (int t = 0; t < n_threads; ++t) g->workers.emplace_back(worker_loop, g);
  *out = g;
  *fd_to_write = p[1];
  return NGSLD_OK;
} catch (...) {
  return NGSLD_ERR_NOMEM;
}

int ngsld_host_gz_close(ngsld_gz *g) {
  if (g == nullptr) return NGSLD_ERR_INVALID;
  // The caller has closed (or fclose'd) the write end -- it owns that descriptor; closing it here by number, after the caller
  // did, could hit an unrelated descriptor that another thread was handed in between.  The reader sees EOF and winds down.
  g->reader.join();
  for (auto &w : g->workers) w.join();
  g->writer.join();
  if (g->fd_read >= 0) ::close(g->fd_read);
  const bool bad = g->failed || ::close(g->fd_out) != 0;
  delete g;
  return bad ? NGSLD_ERR_INVALID : NGSLD_OK;
}

}  // extern "C"
