// Remote user interface: reference src/InterfaceServer.hpp:87-351.
//
// The reference's server speaks `packetcomms` packets over TCP and streams an FFmpeg-encoded preview through `videolib`;
// both are un-vendored submodules (external/packetcomms, external/videolib) and absent here, so the WIRE FORMAT cannot be
// reproduced.  What PathTracerApp's step loop depends on is kept exactly: Status, State, stateChanged() / consumeState()
// / getState() (the update flag is cleared and the NIF request consumed on read, as at :204-209), start() blocking until
// a client is connected (:147-182,228-234), updateProgress / updateSampleRate / sendPreviewImage / startSendingRawImage
// (which drops a request while the previous transfer is still running, :275-279, and sends hdr / step, :284).
//
// Transport of this build: one TCP client on 127.0.0.1:<port>, newline-terminated text.  Client -> server, one command
// per line, named like the reference's packet types (:100-160):
//     env_rotation <degrees> | fov <degrees> | exposure <f> | gamma <f> | interactive_samples <n> | load_nif <path>
//     | stop | detach
// (as in the reference, exposure and gamma do NOT mark the state as updated: tone mapping is host-side, :124-136).
// Server -> client:  "progress <fraction>", "sample_rate <paths/s> <rays/s>",
//     "render_preview <width> <height> <bytes>" followed by <bytes> of raw BGR8 rows (instead of an encoded video packet),
//     "hdr_header <width> <height> <chunks>" followed by <chunks> messages "hdr_packet <index> <bytes>" + one row of RGB
//     float32 each (:286-318).
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "AccumulatedImage.hpp"
#include "AsyncTask.hpp"
#include "logging.hpp"

class InterfaceServer {
public:
  enum class Status { Stop, Restart, Continue, Disconnected };

  struct State {
    float envRotationDegrees = 0.f;
    float exposure = 0.f;
    float gamma = 2.2f;
    float fov = 90.f;   // radians once a client has set it (converted on receipt, as the reference does)
    std::uint32_t interactiveSamples = 8;
    std::string newNif;
    bool stop = false;
    bool detach = false;
  };

  explicit InterfaceServer(int portNumber) : port(portNumber) {}
  virtual ~InterfaceServer() {
    try { sendHdrTask.waitForCompletion(); } catch (...) {}
    stop();
  }

  State consumeState() {
    std::lock_guard<std::mutex> lock(stateMutex);
    State tmp = state;
    stateUpdated = false;   // Clear the update flag.
    state.newNif.clear();   // Clear model load request.
    return tmp;
  }
  State getState() const {
    std::lock_guard<std::mutex> lock(stateMutex);
    return state;
  }
  bool stateChanged() const { return stateUpdated; }

  /// Seed the state a client has not touched yet from the command line (the reference's UI sends every control's
  /// value on connection; a text client may set only what it changes).
  void setInitialState(const State& s) {
    std::lock_guard<std::mutex> lock(stateMutex);
    state = s;
  }

  /// Listen, then block until a client has connected (InterfaceServer.hpp:228-234: start() waits for the Tx/Rx loop).
  void start() {
    stopServer = false;
    serverReady = false;
    stateUpdated = false;
    listenFd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (listenFd < 0) throw std::runtime_error("User interface server: could not create a socket.");
    int one = 1;
    ::setsockopt(listenFd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    addr.sin_port = htons((std::uint16_t)port);
    if (::bind(listenFd, (sockaddr*)&addr, sizeof(addr)) || ::listen(listenFd, 1))
      throw std::runtime_error("User interface server: could not listen on port " + std::to_string(port));
    pt_log::info_("User interface server listening on port {}", port);
    thread.reset(new std::thread(&InterfaceServer::communicate, this));
    while (!serverReady) std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }

  void stop() {
    stopServer = true;
    if (listenFd >= 0) { ::shutdown(listenFd, SHUT_RDWR); ::close(listenFd); listenFd = -1; }
    if (thread) {
      try { thread->join(); } catch (const std::system_error&) { pt_log::error_("User interface server thread could not be joined."); }
      thread.reset();
    }
    std::lock_guard<std::mutex> lock(sendMutex);
    if (clientFd >= 0) { ::close(clientFd); clientFd = -1; }
  }

  void initialiseVideoStream(std::size_t width, std::size_t height) { previewWidth = width; previewHeight = height; }

  void updateProgress(int step, int totalSteps) { sendLine("progress " + std::to_string(step / (float)totalSteps)); }
  void updateSampleRate(float pathRate, float rayRate) {
    sendLine("sample_rate " + std::to_string(pathRate) + " " + std::to_string(rayRate));
  }

  void sendPreviewImage(const Image3<std::uint8_t>& ldr) {
    std::ostringstream h;
    h << "render_preview " << ldr.cols << " " << ldr.rows << " " << ldr.data.size();
    if (!sendBlob(h.str(), ldr.data.data(), ldr.data.size())) pt_log::warn_("Could not send video frame.");
  }

  /// Start transmitting the raw film (hdr / step) row by row on a background task; a request that arrives while the
  /// previous transfer is still in progress is dropped (InterfaceServer.hpp:275-279).
  bool startSendingRawImage(Image3<float>&& rawImage, std::size_t step) {
    if (sendHdrTask.isRunning()) {
      pt_log::debug_("Large data transfer still in progress, dropping request");
      return false;
    }
    sendHdrTask.waitForCompletion();
    hdrImage = std::move(rawImage);
    const float scale = 1.f / step;
    for (auto& v : hdrImage.data) v *= scale;
    if (clientFd < 0) { pt_log::debug_("No client: large data transfer aborted."); return false; }
    const std::size_t chunks = hdrImage.rows;
    sendLine("hdr_header " + std::to_string(hdrImage.cols) + " " + std::to_string(hdrImage.rows) + " " + std::to_string(chunks));
    sendHdrTask.run([this, chunks]() {
      std::vector<float> row(hdrImage.cols * 3);
      for (std::size_t c = 0; c < chunks; ++c) {
        const float* p = hdrImage.ptr(c);
        for (std::size_t x = 0; x < hdrImage.cols; ++x) {   // BGR -> RGB (:299)
          row[3 * x + 0] = p[3 * x + 2]; row[3 * x + 1] = p[3 * x + 1]; row[3 * x + 2] = p[3 * x + 0];
        }
        if (!sendBlob("hdr_packet " + std::to_string(c) + " " + std::to_string(row.size() * 4), row.data(), row.size() * 4)) break;
      }
    });
    return true;
  }

private:
  void communicate() {
    sockaddr_in peer{};
    socklen_t len = sizeof(peer);
    int fd = ::accept(listenFd, (sockaddr*)&peer, &len);
    if (fd < 0) { serverReady = true; return; }   // stop() closed the listener
    { std::lock_guard<std::mutex> lock(sendMutex); clientFd = fd; }
    pt_log::info_("User interface client connected.");
    timeval tv{0, 50000};
    ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    pt_log::info_("User interface server entering Tx/Rx loop.");
    serverReady = true;
    std::string pending;
    char buf[512];
    while (!stopServer) {
      const ssize_t n = ::recv(fd, buf, sizeof(buf), 0);
      if (n == 0) break;                       // client closed: treated as a detach
      if (n < 0) continue;                     // timeout: poll stopServer
      pending.append(buf, (std::size_t)n);
      std::size_t nl;
      while ((nl = pending.find('\n')) != std::string::npos) {
        handle(pending.substr(0, nl));
        pending.erase(0, nl + 1);
      }
      if (pending.size() > kMaxLineBytes) {   // a client that never sends a newline must not grow this buffer for ever
        pt_log::warn_("User interface: dropping {} bytes without a line end", pending.size());
        pending.clear();
      }
    }
    if (!stopServer) {
      std::lock_guard<std::mutex> lock(stateMutex);
      state.detach = true;
      stateUpdated = true;
    }
    pt_log::info_("User interface server Tx/Rx loop exited.");
  }

  // A value the device would refuse (pt_set_render_settings: 1 <= samples <= 65535, 0 < fov < pi) or that did not
  // parse is rejected HERE with a warning and the old state kept: a typo from the client must not abort the render.
  // (The reference's UI sends slider values, which cannot be out of range; a text client can send anything.)
  void handle(const std::string& line) {
    std::istringstream is(line);
    std::string name;
    is >> name;
    auto number = [&](float lo, float hi, bool openInterval, float& out) {
      float v = 0.f;
      std::string rest;
      const bool parsed = static_cast<bool>(is >> v) && std::isfinite(v) && !(is >> rest);
      const bool inRange = openInterval ? (v > lo && v < hi) : (v >= lo && v <= hi);
      if (!parsed || !inRange) {
        pt_log::warn_("User interface: '{}' rejected (want a number in {}{}, {}{}); state unchanged", line, openInterval ? "(" : "[", lo, hi,
                      openInterval ? ")" : "]");
        return false;
      }
      out = v;
      return true;
    };
    std::lock_guard<std::mutex> lock(stateMutex);
    float v = 0.f;
    if (name == "env_rotation") { if (number(-36000.f, 36000.f, false, v)) { state.envRotationDegrees = v; stateUpdated = true; } }
    else if (name == "detach") { state.detach = true; stateUpdated = true; }
    else if (name == "stop") { state.stop = true; stateUpdated = true; }
    else if (name == "exposure") { if (number(-64.f, 64.f, false, v)) state.exposure = v; }        // host-side only: no restart (:124-129)
    else if (name == "gamma") { if (number(0.f, 64.f, true, v)) state.gamma = v; }
    else if (name == "fov") { if (number(0.f, 180.f, true, v)) { state.fov = v * (float)(M_PI / 180.f); stateUpdated = true; } }
    else if (name == "load_nif") { std::getline(is >> std::ws, state.newNif); stateUpdated = true; }
    else if (name == "interactive_samples") {
      if (number(1.f, 65535.f, false, v) && v == std::floor(v)) { state.interactiveSamples = (std::uint32_t)v; stateUpdated = true; }
      else if (v != std::floor(v)) pt_log::warn_("User interface: '{}' rejected (want a whole number); state unchanged", line);
    }
    else if (!name.empty()) pt_log::warn_("User interface: unknown command '{}'", name.substr(0, 64));
  }

  bool sendAll(const void* p, std::size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
      const ssize_t w = ::send(clientFd, c, n, MSG_NOSIGNAL);
      if (w <= 0) return false;
      c += w; n -= (std::size_t)w;
    }
    return true;
  }
  bool sendLine(const std::string& s) {
    std::lock_guard<std::mutex> lock(sendMutex);
    if (clientFd < 0) return false;
    const std::string l = s + "\n";
    return sendAll(l.data(), l.size());
  }
  bool sendBlob(const std::string& header, const void* data, std::size_t bytes) {
    std::lock_guard<std::mutex> lock(sendMutex);
    if (clientFd < 0) return false;
    const std::string l = header + "\n";
    return sendAll(l.data(), l.size()) && sendAll(data, bytes);
  }

  static constexpr std::size_t kMaxLineBytes = 4096;
  int port;
  int listenFd = -1, clientFd = -1;
  std::unique_ptr<std::thread> thread;
  std::atomic<bool> stopServer{false}, serverReady{false}, stateUpdated{false};
  mutable std::mutex stateMutex;
  std::mutex sendMutex;
  State state;
  std::size_t previewWidth = 0, previewHeight = 0;
  Image3<float> hdrImage;
  AsyncTask sendHdrTask;
};
