"""Command-line client for the interactive mode of `ipu_trace --ui-port <port>` (host/InterfaceServer.hpp).

The reference's remote UI speaks packetcomms / videolib, which are not vendored; this build's server speaks newline-terminated
text on 127.0.0.1:<port>.  As in the reference the renderer waits for ONE client, and when that client leaves the server is
gone for the rest of the render -- so a session is one connection, and this script takes the whole session as a list of actions
executed in order:

    ipu_trace --assets <dir> -o out.png -w 720 -h 480 -s 100000 --samples-per-step 64 --ui-port 5000 &
    python scripts/ui_client.py --port 5000 fov=60 env_rotation=120 wait=10 save_preview=a.ppm \\
        load_nif=other/assets.extra wait=10 save_preview=b.ppm save_hdr=b.pfm stop

actions:  env_rotation=<degrees>  fov=<degrees>  exposure=<f>  gamma=<f>  interactive_samples=<n>  load_nif=<assets dir>
          (the reference's packet types, InterfaceServer.hpp:100-160; all but exposure and gamma restart the render)
          wait=<seconds>          keep listening (progress and sample rates are printed as they arrive)
          save_preview=<file>     the latest preview received (raw BGR8 rows) as a binary PPM
          save_hdr=<file>         the latest complete HDR image received (float32 RGB rows) as a PFM
          stop                    end the render: the renderer saves its images and exits
          detach                  leave; the render continues without a user interface
Leaving without `stop` or `detach` is a detach.
"""
import argparse
import socket
import sys
import threading
import time

COMMANDS = ("env_rotation", "fov", "exposure", "gamma", "interactive_samples", "load_nif")


class Session:
    def __init__(self, host, port, quiet=False):
        self.conn = socket.create_connection((host, port), timeout=30)
        self.conn.settimeout(None)
        self.file = self.conn.makefile("rb")
        self.quiet = quiet
        self.preview = None          # (width, height, bytes)
        self.hdr = None              # (width, height, {row: bytes}) of the last COMPLETE image
        self._hdr_next = None
        self.closed = False
        self.lock = threading.Lock()
        self.reader = threading.Thread(target=self._read, daemon=True)
        self.reader.start()

    def _say(self, text):
        if not self.quiet:
            print(text, flush=True)

    def _read(self):
        try:
            while True:
                line = self.file.readline()
                if not line:
                    break
                t = line.decode().split()
                if not t:
                    continue
                if t[0] == "progress":
                    self._say("progress %5.1f %%" % (100.0 * float(t[1])))
                elif t[0] == "sample_rate":
                    self._say("%.4g path-samples/s, %.4g rays/s" % (float(t[1]), float(t[2])))
                elif t[0] == "render_preview":
                    w, h, n = int(t[1]), int(t[2]), int(t[3])
                    data = self.file.read(n)
                    with self.lock:
                        self.preview = (w, h, data)
                elif t[0] == "hdr_header":
                    self._hdr_next = (int(t[1]), int(t[2]), int(t[3]), {})
                elif t[0] == "hdr_packet":
                    data = self.file.read(int(t[2]))
                    if self._hdr_next:
                        w, h, chunks, rows = self._hdr_next
                        rows[int(t[1])] = data
                        if len(rows) == chunks:
                            with self.lock:
                                self.hdr = (w, h, rows)
                            self._hdr_next = None
        except (OSError, ValueError):
            pass
        self.closed = True

    def send(self, line):
        self.conn.sendall((line + "\n").encode())
        self._say("> " + line)

    def save_preview(self, path):
        with self.lock:
            p = self.preview
        if not p:
            self._say("no preview received yet: %s not written" % path)
            return False
        w, h, bgr = p
        rgb = bytearray(bgr)
        rgb[0::3], rgb[2::3] = bgr[2::3], bgr[0::3]
        with open(path, "wb") as out:
            out.write(b"P6\n%d %d\n255\n" % (w, h))
            out.write(bytes(rgb))
        self._say("wrote %s (%d x %d)" % (path, w, h))
        return True

    def save_hdr(self, path):
        with self.lock:
            p = self.hdr
        if not p:
            self._say("no complete HDR image received yet: %s not written" % path)
            return False
        w, h, rows = p
        with open(path, "wb") as out:
            out.write(b"PF\n%d %d\n-1.0\n" % (w, h))            # little-endian float RGB, rows bottom to top
            for row in range(h - 1, -1, -1):
                out.write(rows[row])
        self._say("wrote %s (%d x %d)" % (path, w, h))
        return True

    def close(self):
        try:
            self.conn.shutdown(socket.SHUT_RDWR)
        except OSError:
            pass
        self.file.close()
        self.conn.close()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--connect-timeout", type=float, default=60.0, help="seconds to keep trying while the renderer starts up")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("actions", nargs="*", help="see above")
    args = ap.parse_args()
    for a in args.actions:
        name = a.split("=", 1)[0]
        if name not in COMMANDS + ("wait", "save_preview", "save_hdr", "stop", "detach") or (name not in ("stop", "detach") and "=" not in a):
            ap.error("unknown action '%s'" % a)

    s, t0 = None, time.time()
    while s is None:
        try:
            s = Session(args.host, args.port, args.quiet)
        except OSError:
            if time.time() - t0 > args.connect_timeout:
                raise SystemExit("no user-interface server on %s:%d" % (args.host, args.port))
            time.sleep(0.2)
    ok = True
    for a in args.actions:
        name, _, value = a.partition("=")
        if name in COMMANDS:
            s.send("%s %s" % (name, value))
        elif name == "wait":
            end = time.time() + float(value)
            while time.time() < end and not s.closed:
                time.sleep(0.05)
        elif name == "save_preview":
            ok = s.save_preview(value) and ok
        elif name == "save_hdr":
            ok = s.save_hdr(value) and ok
        else:
            s.send(name)
    s.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
