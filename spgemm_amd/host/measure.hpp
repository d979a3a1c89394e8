// START/END measurement points and per-run byte counts, written as the reference's
// measurements_<rank>.csv: header "func,duration,bytes", one "bytes,0,<n>" row per timed run,
// then one "<func>,<ns>,0" row per interval (reference src/measure.cpp:103-115).
// One instance per rank (the ranks of this driver live in one process).
#pragma once

#include <chrono>
#include <fstream>
#include <map>
#include <string>
#include <vector>

namespace measure {

enum class MeasurementEvent { START, END };

// phase names of the reference (include/measure.hpp:16-30) plus the device phases of this engine
inline const std::string shuffle = "shuffle", partition = "partition", bitmaps = "bitmaps", gemm = "gemm",
                         global = "global", mult = "mult", filter = "filter", deserialize = "deserialize",
                         wait_all = "wait_all", read_triplets = "read_triplets", build_csr = "build_csr",
                         symbolic = "symbolic", numeric = "numeric", analyze = "analyze";

class Measure {
  struct Point { std::string func; MeasurementEvent ev; std::chrono::high_resolution_clock::time_point t; };
  std::vector<Point> points;
  std::vector<std::pair<std::string, double>> extra; // intervals measured elsewhere (device events), ns
  std::vector<size_t> bytes_runs;
  size_t bytes_ = 0;

public:
  static Measure *get_instance(int rank) {
    static std::map<int, Measure> inst;
    return &inst[rank];
  }
  void track(const std::string &func, MeasurementEvent ev) { points.push_back({func, ev, std::chrono::high_resolution_clock::now()}); }
  void add_interval_ns(const std::string &func, double ns) { extra.push_back({func, ns}); }
  void track_bytes(size_t b) { bytes_ += b; }
  void flush_bytes() { bytes_runs.push_back(bytes_); bytes_ = 0; }
  void reset_bytes() { bytes_ = 0; }
  void reset() { points.clear(); extra.clear(); bytes_ = 0; }

  void save(const std::string &path) {
    std::ofstream csv(path);
    csv << "func,duration,bytes" << std::endl;
    for (auto b : bytes_runs) csv << "bytes,0," << b << std::endl;
    std::vector<bool> used(points.size(), false);
    for (size_t i = 0; i < points.size(); ++i) {
      if (used[i] || points[i].ev != MeasurementEvent::START) continue;
      for (size_t j = i + 1; j < points.size(); ++j)
        if (!used[j] && points[j].ev == MeasurementEvent::END && points[j].func == points[i].func) {
          auto ns = std::chrono::duration_cast<std::chrono::nanoseconds>(points[j].t - points[i].t).count();
          csv << points[i].func << "," << ns << ",0" << std::endl;
          used[i] = used[j] = true;
          break;
        }
    }
    for (auto &[f, ns] : extra) csv << f << "," << (long long)ns << ",0" << std::endl;
  }
};

} // namespace measure

#define measure_point(rank, name, event) measure::Measure::get_instance(rank)->track(name, event)
