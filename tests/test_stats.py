"""The render report: the reference's own tests of stats.rs, on crust-render_amd/stats.py."""
import importlib

stats = importlib.import_module("crust-render_amd.stats")
RenderStats, PrimitiveCounts, SceneCounters = stats.RenderStats, stats.PrimitiveCounts, stats.SceneCounters


def test_total_skips_nested_phases():  # stats.rs:568-577
    s = RenderStats()
    s.record("Parse", 0, 10)
    s.record("Open stage", 1, 4)
    s.record("Traverse", 1, 6)
    s.record("Render", 0, 30)
    assert s.total() == 40  # 10 + 30, not 10 + 4 + 6 + 30


def test_report_lists_every_phase_in_both_views():  # stats.rs:579-594
    s = RenderStats()
    s.record("Parse USD stage", 0, 2)
    s.record("Load assets", 1, 0.5)
    s.record("Trace paths", 0, 8)
    out = s.report()
    assert "Profile by execution tree" in out and "Profile by time" in out
    assert out.count("Load assets") == 2 and out.count("Trace paths") == 2 and out.count("Parse USD stage") == 2


def test_percentages_are_relative_to_top_level_total():  # stats.rs:596-604
    s = RenderStats()
    s.record("A", 0, 1)
    s.record("B", 0, 3)
    out = s.report()
    assert "25.0%" in out and "75.0%" in out


def test_zero_counts_are_omitted_from_the_breakdown():  # stats.rs:606-621
    s = RenderStats()
    s.scene = SceneCounters(top_level=PrimitiveCounts(triangles=12))
    out = s.report()
    assert "triangles" in out and "spheres" not in out


def test_unique_breakdown_is_shown_only_when_it_differs():  # stats.rs:623-660
    flat = PrimitiveCounts(triangles=3)
    same = RenderStats()
    same.scene = SceneCounters(top_level=flat, unique=PrimitiveCounts(triangles=3))
    assert "primitives in memory" not in same.report()
    inst = RenderStats()
    inst.scene = SceneCounters(top_level=PrimitiveCounts(instances=2),
                               unique=PrimitiveCounts(instances=2, cubic_curve_spans=900))
    out = inst.report()
    assert "primitives in memory" in out and "900" in out


def test_primitive_counts_total_every_kind():  # stats.rs:662-672
    assert PrimitiveCounts(1, 2, 3, 4, 5).total() == 15


def test_thousands_separates_groups_of_three():  # stats.rs:674-679
    assert stats.thousands(7) == "7" and stats.thousands(1234) == "1 234" and stats.thousands(1234567) == "1 234 567"


def test_human_bytes_scales_to_gibibytes():  # stats.rs:681-685
    assert stats.human_bytes(512) == "512 B" and stats.human_bytes(2 * 1024 ** 3) == "2.00 GiB"


def test_human_duration_forms():  # stats.rs:329-338
    assert stats.human_duration(1.5) == "  1.500s"
    assert stats.human_duration(75.25) == "01:15.2" or stats.human_duration(75.25) == "01:15.3"
    assert stats.human_duration(3600.0) == "60:00.0"


def test_ray_statistics_block_and_throughput():  # stats.rs:101-160, :446-500
    s = RenderStats()
    s.rays = stats.RayStats.of(dict(camera_rays=1000, closest_hit=2500, shadow_rays=500, vertices=1800, rr_tested=200,
                                    rr_killed=50, ended_escaped=700, ended_depth=3))
    assert s.rays.total_rays() == 3000 and s.rays.mean_path_length() == 1.8 and s.rays.rr_kill_rate() == 0.25
    s.record("Parse", 0, 1.0)
    s.record("Render", 0, 0.001)
    out = s.report()
    assert "Ray Statistics" in out and "3.00 Mray/s" in out and "50 of 200 (25.0%)" in out
    assert "mean path length             1.80" in out
    merged = stats.RayStats()
    merged.merge(s.rays)
    merged.merge(s.rays)
    assert merged.closest_hit == 5000 and merged.ended_depth == 6
    assert "Ray Statistics" not in RenderStats().report()  # empty counters: the block is omitted


def test_phase_context_manager_records_in_order():
    s = RenderStats()
    with s.phase("Load"):
        with s.phase("Parse", 1):
            pass
    with s.phase("Render"):
        pass
    assert [(p.name, p.depth) for p in s.phases] == [("Parse", 1), ("Load", 0), ("Render", 0)]
    assert all(p.rss_end is None or p.rss_end > 0 for p in s.phases)
