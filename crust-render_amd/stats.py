"""Render statistics and the per-phase profile report, in the reference's report layout so the two
renderers' logs can be read side by side (crates/crust-core/src/stats.rs):

  RenderStats{phases, scene, image, rays}   stats.rs:214-264   record / record_at / total / report
  Phase, MemorySample                        stats.rs:21-45
  PrimitiveCounts, SceneCounters             stats.rs:47-99
  ImageCounters                              stats.rs:170-190
  report text                                stats.rs:340-562

RayStats itself is the C-ABI's CrtRayStats (stats.rs:101-160): its derived figures are the helpers below.
`for_render()` fills a report from a committed scene and a finished render of this package's Renderer.
"""
import dataclasses
import time
from contextlib import contextmanager

_KINDS = ("triangles", "spheres", "curve_segments", "cubic_curve_spans", "instances")
_RAY_FIELDS = ("camera_rays", "closest_hit", "shadow_rays", "vertices", "rr_tested", "rr_killed", "ended_escaped",
               "ended_depth")


def _proc_status_bytes(field):  # stats.rs:266-277
    try:
        with open("/proc/self/status") as f:
            for line in f:
                if line.startswith(field):
                    return int(line[len(field):].split()[0]) * 1024
    except (OSError, ValueError, IndexError):
        pass
    return None


def peak_memory_bytes():  # stats.rs:279-288 (VmHWM)
    return _proc_status_bytes("VmHWM:")


def current_memory_bytes():  # stats.rs:290-299 (VmRSS)
    return _proc_status_bytes("VmRSS:")


def thousands(n):  # stats.rs:301-311: groups of three separated by a space
    return f"{int(n):,}".replace(",", " ")


def human_bytes(b):  # stats.rs:313-327
    units = ("B", "KiB", "MiB", "GiB", "TiB")
    v, unit = float(b), 0
    while v >= 1024.0 and unit < len(units) - 1:
        v /= 1024.0
        unit += 1
    return f"{int(b)} {units[0]}" if unit == 0 else f"{v:.2f} {units[unit]}"


def human_duration(secs):  # stats.rs:329-338
    if secs >= 60.0:
        mins = secs // 60.0
        return f"{mins:02.0f}:{secs - mins * 60.0:04.1f}"
    return f"{secs:7.3f}s"


@dataclasses.dataclass
class MemorySample:  # stats.rs:31-45
    rss: int = None
    peak: int = None

    @staticmethod
    def now():
        return MemorySample(current_memory_bytes(), peak_memory_bytes())


@dataclasses.dataclass
class Phase:  # stats.rs:21-29: depth 0 = top level; nested phases are counted in their parent
    name: str
    depth: int
    duration: float  # seconds
    rss_end: int = None
    peak_end: int = None


@dataclasses.dataclass
class PrimitiveCounts:  # stats.rs:47-67
    triangles: int = 0
    spheres: int = 0
    curve_segments: int = 0
    cubic_curve_spans: int = 0
    instances: int = 0

    def total(self):
        return sum(getattr(self, k) for k in _KINDS)

    def is_empty(self):
        return self.total() == 0


@dataclasses.dataclass
class SceneCounters:  # stats.rs:83-99; footprint = MemoryFootprint as the dict Scene.memory_footprint() returns
    geometries: int = 0
    top_level: PrimitiveCounts = dataclasses.field(default_factory=PrimitiveCounts)
    unique: PrimitiveCounts = dataclasses.field(default_factory=PrimitiveCounts)
    lights: int = 0
    volumes: int = 0
    footprint: dict = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class ImageCounters:  # stats.rs:170-190
    width: int = 0
    height: int = 0
    samples_per_pixel: int = 0
    max_depth: int = 0


@dataclasses.dataclass
class RayStats:  # stats.rs:101-160
    camera_rays: int = 0
    closest_hit: int = 0
    shadow_rays: int = 0
    vertices: int = 0
    rr_tested: int = 0
    rr_killed: int = 0
    ended_escaped: int = 0
    ended_depth: int = 0

    @staticmethod
    def of(counters):
        """From anything carrying the eight counters (CrtRayStats, the oracle's RayStats, a dict)."""
        get = counters.get if isinstance(counters, dict) else lambda k: getattr(counters, k)
        return RayStats(**{k: int(get(k)) for k in _RAY_FIELDS})

    def total_rays(self):
        return self.closest_hit + self.shadow_rays

    def mean_path_length(self):
        return self.vertices / self.camera_rays if self.camera_rays else 0.0

    def rr_kill_rate(self):
        return self.rr_killed / self.rr_tested if self.rr_tested else 0.0

    def merge(self, o):
        for k in _RAY_FIELDS:
            setattr(self, k, getattr(self, k) + getattr(o, k))

    def is_empty(self):
        return all(getattr(self, k) == 0 for k in _RAY_FIELDS)


class RenderStats:  # stats.rs:214-264
    def __init__(self):
        self.phases = []
        self.scene = SceneCounters()
        self.image = ImageCounters()
        self.rays = RayStats()

    def record(self, name, depth, duration):
        """A completed phase, sampling memory now (stats.rs:232-234)."""
        self.record_at(name, depth, duration, MemorySample.now())

    def record_at(self, name, depth, duration, mem):  # stats.rs:238-252
        self.phases.append(Phase(name, int(depth), float(duration), mem.rss, mem.peak))

    @contextmanager
    def phase(self, name, depth=0):
        """`with stats.phase("Render"):` — times the block and records it when it ends. A parent is recorded
        after its children, as the reference's callers do; the report lists phases in recording order."""
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.record(name, depth, time.perf_counter() - t0)

    def total(self):  # stats.rs:254-260: nested phases are already inside their parents
        return sum(p.duration for p in self.phases if p.depth == 0)

    def report(self):  # stats.rs:262-264, :340-562
        WIDTH, NAME = 84, 36
        rule = "-" * WIDTH
        total = self.total()

        def pct(d):
            return 100.0 * d / total if total > 0 else 0.0

        out = [rule, "Render Statistics", rule]
        img = self.image
        if img.width > 0 and img.height > 0:
            out.append(f"  {'resolution':<28} {img.width}x{img.height}")
            out.append(f"  {'samples per pixel':<28} {img.samples_per_pixel}")
            out.append(f"  {'max path depth':<28} {img.max_depth}")
        s = self.scene
        out.append(f"  {'geometries':<28} {thousands(s.geometries)}")

        def breakdown(title, c):  # zero counts are skipped
            out.append(f"  {title:<28} {thousands(c.total())}")
            for label, k in (("triangles", "triangles"), ("spheres", "spheres"), ("curve segments", "curve_segments"),
                             ("cubic curve spans", "cubic_curve_spans"), ("instances", "instances")):
                if getattr(c, k) > 0:
                    out.append(f"    {label:<26} {thousands(getattr(c, k))}")

        breakdown("top-level BVH primitives", s.top_level)
        if not s.unique.is_empty() and s.unique != s.top_level:  # only when instancing made the two differ
            breakdown("primitives in memory", s.unique)
        out.append(f"  {'lights':<28} {thousands(s.lights)}")
        if s.volumes > 0:
            out.append(f"  {'volume regions':<28} {thousands(s.volumes)}")
        fp = s.footprint or {}
        if sum(fp.values()) > 0:
            out.append(f"  {'kernel memory':<28} {human_bytes(sum(fp.values()))}")
            for label, k in (("primitive nodes", "prim_nodes"), ("boxed primitives", "boxed_prims"),
                             ("BVH nodes", "bvh_nodes"), ("triangle packets", "packets"), ("leaf indices", "indices"),
                             ("leaves", "leaves")):
                if fp.get(k, 0) > 0:
                    out.append(f"    {label:<26} {human_bytes(fp[k])}")
        peak = peak_memory_bytes()
        if peak is not None:
            out.append(f"  {'peak memory (RSS)':<28} {human_bytes(peak)}")

        r = self.rays
        if not r.is_empty():
            out += [rule, "Ray Statistics", rule]
            out.append(f"  {'camera rays':<28} {thousands(r.camera_rays)}")
            out.append(f"  {'closest-hit queries':<28} {thousands(r.closest_hit)}")
            out.append(f"  {'shadow rays':<28} {thousands(r.shadow_rays)}")
            out.append(f"  {'total ray queries':<28} {thousands(r.total_rays())}")
            out.append(f"  {'vertices shaded':<28} {thousands(r.vertices)}")
            out.append(f"  {'mean path length':<28} {r.mean_path_length():.2f}")
            render = next((p for p in self.phases if p.depth == 0 and p.name == "Render"), None)
            if render is not None and render.duration > 0:  # throughput over the render phase alone
                rps = r.total_rays() / render.duration
                v, unit = (rps / 1e6, "Mray/s") if rps >= 1e6 else (rps / 1e3, "Kray/s") if rps >= 1e3 else (rps, "ray/s")
                out.append(f"  {'throughput':<28} {v:.2f} {unit}")
                out.append(f"  {'mean time per ray query':<28} {1e6 * render.duration / max(r.total_rays(), 1):.2f} us")
            out.append(f"  {'roulette kills':<28} {thousands(r.rr_killed)} of {thousands(r.rr_tested)} "
                       f"({100.0 * r.rr_kill_rate():.1f}%)")
            out.append(f"  {'paths ended: escaped':<28} {thousands(r.ended_escaped)}")
            out.append(f"  {'paths ended: depth cap':<28} {thousands(r.ended_depth)}")

        if not self.phases:
            return "\n".join(out) + "\n"

        def mem(b):
            return human_bytes(b) if b is not None else ""

        out += [rule, "Profile by execution tree", f"{'':<{NAME}} {'time':>9}  {'%':>5}  {'rss':>9} {'peak':>9}", rule]
        for p in self.phases:
            indent = "  " * (1 + p.depth)
            w = max(NAME - len(indent), 0)
            out.append(f"{indent}{p.name:<{w}} {human_duration(p.duration):>9}  {pct(p.duration):>5.1f}%  "
                       f"{mem(p.rss_end):>9} {mem(p.peak_end):>9}")
        out.append(f"  {'total':<{NAME - 2}} {human_duration(total):>9}")
        out += [rule, "Profile by time (* = nested, counted in its parent)", rule]
        for p in sorted(self.phases, key=lambda p: -p.duration):  # stable: equal durations keep recording order
            marker = "*" if p.depth > 0 else " "
            out.append(f"  {marker}{p.name:<{NAME - 3}} {human_duration(p.duration):>9}  {pct(p.duration):>5.1f}%")
        out.append(rule)
        return "\n".join(out)

    def __str__(self):
        return self.report()


def for_render(scene, renderer, spp, stats=None):
    """Fills the scene / image / ray blocks of a report from a committed Scene and a Renderer that has rendered
    `spp` samples (what main.rs assembles around Renderer::render, main.rs:560-640). Phases are the caller's:
    wrap the load / commit / render steps in `stats.phase(...)`."""
    st = stats or RenderStats()
    st.scene = SceneCounters(geometries=scene.geometry_count(),
                             top_level=PrimitiveCounts(**scene.primitive_breakdown()),
                             unique=PrimitiveCounts(**scene.unique_primitive_breakdown()),
                             lights=renderer.n_lights, volumes=0, footprint=scene.memory_footprint())
    s = renderer.settings
    st.image = ImageCounters(s.width, s.height, int(spp), s.max_depth)
    st.rays = RayStats.of(renderer.stats())
    return st
