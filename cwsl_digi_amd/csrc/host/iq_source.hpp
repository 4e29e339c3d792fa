// iq_source.hpp -- where a receiver's IQ blocks come from on Linux.
//
// The reference reads one block of BlockInSamples interleaved complex<float> per wake-up from CWSL's Win32 shared
// memory, whose 12-byte header carries {SampleRate, BlockInSamples, L0} (SharedMemory.h:10-21, Receiver.hpp:86-88,
// :215-249).  Here the same stream comes from
//   file   a "band file": that 12-byte header followed by the blocks, or headerless raw complex64 with the three
//          numbers given on the command line ("-" = stdin)
//   udp    one datagram = a whole number of SSBD input quanta of raw complex64 (fs/block/lo from the command line);
//          bind=ADDR picks the local address (default 127.0.0.1), idle=S ends the stream after S silent seconds
//          (default 0: keep waiting, as the reference's WaitForNewData loop does); malformed datagrams are counted
//          and dropped, never fatal
#pragma once
#include <arpa/inet.h>
#include <cerrno>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace cwslg {
namespace host {

struct BandHeader { int32_t SampleRate, BlockInSamples, L0; };   // SharedMemory.h:10-21

struct RxSpec {
    std::string kind, path;        // "file" | "udp"
    int port = 0;
    uint32_t fs = 0, block = 0;
    int64_t lo = 0;
    bool header = false;           // the file starts with a BandHeader
    std::string bind_addr = "127.0.0.1";
    double idle_s = 0;             // udp: 0 = wait for data for ever
};

// "file=PATH[,fs=..,block=..,lo=..][,header=1]" or "udp=PORT,fs=..,block=..,lo=.."
inline bool parse_rx_spec(const std::string &arg, RxSpec &r, std::string &err)
{
    size_t pos = 0;
    while (pos < arg.size()) {
        size_t comma = arg.find(',', pos);
        if (comma == std::string::npos) comma = arg.size();
        const std::string item = arg.substr(pos, comma - pos);
        const size_t eq = item.find('=');
        if (eq == std::string::npos) { err = "bad --rx item: " + item; return false; }
        const std::string k = item.substr(0, eq), v = item.substr(eq + 1);
        if (k == "file") { r.kind = "file"; r.path = v; }
        else if (k == "udp") { r.kind = "udp"; r.port = std::atoi(v.c_str()); }
        else if (k == "fs") r.fs = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (k == "block") r.block = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (k == "lo") r.lo = std::strtoll(v.c_str(), nullptr, 10);
        else if (k == "header") r.header = v != "0";
        else if (k == "bind") r.bind_addr = v;
        else if (k == "idle") r.idle_s = std::atof(v.c_str());
        else { err = "unknown --rx key: " + k; return false; }
        pos = comma + 1;
    }
    if (r.kind.empty()) { err = "--rx needs file= or udp="; return false; }
    if (!r.header && (r.fs == 0 || r.block == 0)) { err = "--rx without header=1 needs fs= and block="; return false; }
    return true;
}

class IqSource {
public:
    ~IqSource() { close(); }
    bool open(RxSpec &r, std::string &err)
    {
        spec_ = &r;
        if (r.kind == "file") {
            f_ = (r.path == "-") ? stdin : std::fopen(r.path.c_str(), "rb");
            if (!f_) { err = "cannot open " + r.path; return false; }
            if (r.header) {
                BandHeader h;
                if (std::fread(&h, sizeof h, 1, f_) != 1 || h.SampleRate <= 0 || h.BlockInSamples <= 0) { err = "bad band header in " + r.path; return false; }
                r.fs = (uint32_t)h.SampleRate; r.block = (uint32_t)h.BlockInSamples; r.lo = h.L0;
            }
            return true;
        }
        sock_ = ::socket(AF_INET, SOCK_DGRAM, 0);
        if (sock_ < 0) { err = "socket() failed"; return false; }
        int big = 8 << 20;
        ::setsockopt(sock_, SOL_SOCKET, SO_RCVBUF, &big, sizeof big);
        sockaddr_in a{};
        a.sin_family = AF_INET; a.sin_port = htons((uint16_t)r.port);
        if (::inet_pton(AF_INET, r.bind_addr.c_str(), &a.sin_addr) != 1) { err = "bad bind address " + r.bind_addr; return false; }
        if (::bind(sock_, (sockaddr *)&a, sizeof a) != 0) { err = "cannot bind udp " + r.bind_addr + ":" + std::to_string(r.port); return false; }
        timeval tv{1, 0};                                     // SM.WaitForNewData(1000): wake up once a second (Receiver.hpp:215-221)
        ::setsockopt(sock_, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        return true;
    }
    // one Receiver block (file) or one datagram (udp) of interleaved complex64; returns complex samples read, 0 at end
    uint32_t read(std::vector<std::complex<float>> &buf)
    {
        if (f_) {
            buf.resize(spec_->block);
            const size_t n = std::fread(buf.data(), sizeof(buf[0]), spec_->block, f_);
            return (n == spec_->block) ? spec_->block : 0;   // a trailing partial block is dropped, as a short SM.Read would be
        }
        // a datagram must be a whole number of SSBD input quanta (2*Fs/B = Fs/3000 complex samples): anything else --
        // a torn write, a stray sender -- is dropped and counted; cwslg_push_iq would reject it (CWSLG_ERR_BLOCK)
        const size_t quantum = spec_->fs / 3000;
        double silent = 0;
        for (;;) {
            buf.resize(8192);
            const ssize_t n = ::recv(sock_, buf.data(), buf.size() * sizeof(buf[0]), 0);
            if (n < 0) {
                if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) {      // nothing yet: keep waiting
                    silent += 1.0;
                    if (spec_->idle_s > 0 && silent >= spec_->idle_s) return 0;
                    continue;
                }
                return 0;
            }
            silent = 0;
            const size_t ns = (size_t)n / sizeof(buf[0]);
            if (n == 0 || (size_t)n % sizeof(buf[0]) != 0 || quantum == 0 || ns % quantum != 0) { ++bad_datagrams; continue; }
            return (uint32_t)ns;
        }
    }
    uint64_t bad_datagrams = 0;
    void close()
    {
        if (f_ && f_ != stdin) std::fclose(f_);
        f_ = nullptr;
        if (sock_ >= 0) ::close(sock_);
        sock_ = -1;
    }
private:
    RxSpec *spec_ = nullptr;
    FILE *f_ = nullptr;
    int sock_ = -1;
};

}  // namespace host
}  // namespace cwslg
