// shk_count — sharkmer's counting-related command line over libshk:
//   shk_count -k 21 --chunks 10 --histo-max 10000 -m 1000000 -s sample -o outdir/ reads.fastq.gz …
// Flags mirror the reference's clap definitions (src/cli.rs:165-232): -k (default 19), --chunks
// (0), --histo-max (10000), -m/--max-reads, -s/--sample, -o/--outdir ("./"), -t/--threads
// (accepted; counting was single-threaded in the reference and is on the GPU here),
// --validate-every (0).  Exit status and "Error: …" on stderr follow anyhow's main().
#include "../../include/shk.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool parse_u64(const char *s, unsigned long long *out) {
  if (!s || !*s) return false;
  char *end = nullptr;
  *out = strtoull(s, &end, 10);
  return end && *end == 0;
}

int main(int argc, char **argv) {
  bool gzip_all = false;
  unsigned long long k = 19, chunks = 0, histo_max = 10000, max_reads = 0, validate_every = 0, threads = 1,
                     device = 0, hint = 0;
  const char *sample = nullptr, *outdir = "./";
  std::vector<int32_t> devices;  // --devices 0,1,2,3: one multi-device context (not a reference flag)
  std::vector<const char *> inputs;
  std::string command;
  for (int i = 0; i < argc; ++i) {
    if (i) command += ' ';
    command += argv[i];
  }
  auto need = [&](int &i, const char *flag) -> const char * {
    if (i + 1 >= argc) {
      fprintf(stderr, "error: a value is required for '%s' but none was supplied\n", flag);
      exit(2);
    }
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    const char *val = nullptr;
    auto eq = a.find('=');
    std::string key = a, inl;
    if (a.rfind("--", 0) == 0 && eq != std::string::npos) {
      key = a.substr(0, eq);
      inl = a.substr(eq + 1);
      val = inl.c_str();
    }
    auto num = [&](unsigned long long *dst) {
      const char *v = val ? val : need(i, key.c_str());
      if (!parse_u64(v, dst)) {
        fprintf(stderr, "error: invalid value '%s' for '%s'\n", v, key.c_str());
        exit(2);
      }
    };
    if (key == "-k") num(&k);
    else if (key == "--chunks") num(&chunks);
    else if (key == "--histo-max") num(&histo_max);
    else if (key == "-m" || key == "--max-reads") num(&max_reads);
    else if (key == "--validate-every") num(&validate_every);
    else if (key == "--gzip-all-members") gzip_all = true;  // (not a sharkmer flag: sharkmer reads a gzip file's first member)
    else if (key == "-t" || key == "--threads") num(&threads);
    else if (key == "--device") num(&device);
    else if (key == "--capacity-hint") num(&hint);
    else if (key == "--devices") {
      const char *v = val ? val : need(i, key.c_str());
      for (const char *p = v; *p;) {
        char *end = nullptr;
        devices.push_back((int32_t)strtol(p, &end, 10));
        if (end == p) {
          fprintf(stderr, "error: invalid value '%s' for '--devices'\n", v);
          return 2;
        }
        p = *end == ',' ? end + 1 : end;
      }
    }
    else if (key == "-s" || key == "--sample") sample = val ? argv[i] + eq + 1 : need(i, key.c_str());
    else if (key == "-o" || key == "--outdir") outdir = val ? argv[i] + eq + 1 : need(i, key.c_str());
    else if (key == "-h" || key == "--help") {
      printf("Usage: shk_count [-k K] [--chunks N] [--histo-max M] [-m READS] -s SAMPLE [-o OUTDIR] "
             "[--validate-every N] [--gzip-all-members] [FASTQ[.gz] ...]\n");
      return 0;
    } else if (a.size() > 1 && a[0] == '-' && a != "-") {
      fprintf(stderr, "error: unexpected argument '%s' found\n", a.c_str());
      return 2;
    } else {
      inputs.push_back(argv[i]);
    }
  }
  shk_run_config rc{};
  rc.inputs = inputs.data();
  rc.n_inputs = (uint32_t)inputs.size();
  rc.k = (uint32_t)k;
  rc.chunks = (uint32_t)chunks;
  rc.device = (int32_t)device;
  rc.histo_max = histo_max;
  rc.max_reads = max_reads;
  rc.validate_every = validate_every;
  rc.fastq_flags = gzip_all ? SHK_FASTQ_GZIP_ALL_MEMBERS : 0;
  rc.sample = sample;
  rc.outdir = outdir;
  rc.command = command.c_str();
  rc.table_capacity_hint = hint;
  rc.n_devices = (uint32_t)devices.size();
  rc.device_ids = devices.empty() ? nullptr : devices.data();
  shk_run_stats st{};
  int rcode = shk_run_files(&rc, &st);
  if (rcode != SHK_OK) {
    fprintf(stderr, "Error: %s\n", shk_run_error());
    return 1;
  }
  fprintf(stderr, "%llu reads, %llu kmers\n", (unsigned long long)st.n_reads_read, (unsigned long long)st.n_kmers);
  return 0;
}
