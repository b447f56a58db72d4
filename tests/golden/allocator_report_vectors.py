"""Hand-derived known-answer vectors for the host-allocator job's report math.

PARITY UNPINNED BY THE REFERENCE: units/host_allocator.go:250-334 and setTargetAndTerminate (:393-424) have no test in
the reference that holds expected values, so these vectors were worked out by hand from the Go source, one per branch,
with the arithmetic written next to each (T = MaxDurationThreshold, MIN = 60e9 ns, HOUR = 3600e9 ns). They pin the oracle,
the host-object restatement and the HIP kernel to ONE reading of the code -- the author's; a misreading shared by all
three would not be caught here.

What each statement computes, line by line of the reference (units/host_allocator.go), and where each of the three
statements does it -- oracle/evg_oracle.cpp (evg_oracle_allocator_report), tests/host_restatements.py (HostAllocatorReport, the
host-object form; allocator_report_rows, the vectorised form used for the 10^6-row fuzz) and csrc/evg_sched.hip (k_allocator_report):

  Go line   quantity                                               formula (all int64 unless said)
  :267-277  sums over TaskGroupInfos with Name != ""               over, durOver, dur, free, required  (CountWaitOverThreshold is summed
                                                                   at :269 and only logged: not part of any decision)
  :280      correctedExpectedDuration                              info.ExpectedDuration - dur
  :282      correctedDurationOverThreshold                         info.DurationOverThreshold - durOver
  :284      scheduledDuration                                      :280 - :282
  :286      durationOverThreshNoTaskGroups                         info.CountDurationOverThreshold - over
  :289      correctedHostsSpawned                                  len(hostsSpawned) - required
  :291      hostsAvail                                             (nHostsFree - free) + :289 - :286
  :299-301  scheduledDuration <= 0                                 timeToEmpty = timeToEmptyNoSpawns = 0
  :304      hostsAvailNoSpawns                                     hostsAvail - :289
  :305-308  hostsAvail <= 0                                        both = 2532000 * time.Hour (maxPossibleHours)
  :309-311  hostsAvailNoSpawns <= 0                                timeToEmpty = scheduled / hostsAvail (Go int64 division), the other = max
  :312-315  otherwise                                              scheduled / hostsAvail, scheduled / hostsAvailNoSpawns
  :319      hostQueueRatio   (float32)                             float32(timeToEmpty) / float32(MaxDurationThreshold): x/0 = +Inf, 0/0 = NaN
  :321      noSpawnsRatio    (float32)                             float32(timeToEmptyNoSpawns) / float32(MaxDurationThreshold)
  :324-333  the drawdown gate                                      terminate rule && spawnable provider && ratio < float32(.25) && len(upHosts) > 0
                                                                   && !hourly billing; the three host-side facts travel as drawdown_allowed
  :395-396  hostQueueRatio == 0                                    killableHosts = numUpHosts, newCapTarget stays 0
  :397-399  otherwise                                              killableHosts = int(float32(numUpHosts) * (1 - hostQueueRatio)) (float32 product,
                                                                   truncated), newCapTarget = numUpHosts - killableHosts
  :402-404  newCapTarget < MinimumHosts                            newCapTarget = MinimumHosts
  :407      killableHosts > 0                                      a drawdown job with NewCapTarget is enqueued (drawdown = 1)

The fuzz tests (tests/test_allocator_report.py: test_parity_unpinned_*) aim rows at the float32 neighbourhood of the 0.25 line, at both
maxPossibleHours branches, at scheduledDuration <= 0 and at a zero threshold, and require the three statements to agree bit for bit
(float32 included) -- 300,000 rows on the CPU, 1,000,000 through the HIP kernel. That is agreement of readings, not a pin.

Each vector: (name, lines, info, groups, hosts_spawned, n_hosts_free, n_up_hosts, minimum_hosts, drawdown_allowed, want)
  info   = (ExpectedDuration, DurationOverThreshold, CountDurationOverThreshold, MaxDurationThreshold)
  groups = [(Name, ExpectedDuration, DurationOverThreshold, CountDurationOverThreshold, CountFree, CountRequired)]
  want   = (timeToEmpty, timeToEmptyNoSpawns, hostsAvail, drawdown, NewCapTarget, killableHosts); the two float32 ratios
           are float32(timeToEmpty) / float32(T) and float32(timeToEmptyNoSpawns) / float32(T) (:319-321) -- the test forms
           them with numpy.float32 from the integers below.
"""
MIN = 60 * 10**9
HOUR = 60 * MIN
MAX = 2532000 * HOUR  # :305 maxPossibleHours
T30 = 30 * MIN

VECTORS = [
    # scheduledDuration = (40m - 0) - (40m - 0) = 0 -> both times 0 (:299-301). hostsAvail = (3 - 0) + (2 - 0) - (1 - 0) = 4
    # (:289-291). ratio = 0 < 0.25, 5 hosts up, drawdown allowed: hostQueueRatio == 0 -> killable = 5, target stays 0 (:395-396),
    # raised to MinimumHosts = 1 (:402-404); killable > 0 -> drawdown (:407).
    ("nothing short is queued; ratio 0 kills every host down to the minimum", "299-301,395-396,402-407",
     (40 * MIN, 40 * MIN, 1, T30), [], 2, 3, 5, 1, True, (0, 0, 4, True, 1, 5)),
    # scheduled = 100m > 0; hostsAvail = 0 + 0 - 0 = 0 -> both = 2532000 h (:306-308). ratio ~ 5e6: no drawdown.
    ("no host available: both estimates are the maximum duration", "306-308",
     (100 * MIN, 0, 0, T30), [], 0, 0, 7, 0, True, (MAX, MAX, 0, False, 0, 0)),
    # scheduled = 60m; spawned 3, free 0: hostsAvail = 0 + 3 - 0 = 3, hostsAvailNoSpawns = 3 - 3 = 0 (:304) ->
    # timeToEmpty = 60m / 3 = 20m, no-spawns = max (:309-311). ratio = 20m / 30m = 0.667 >= 0.25: no drawdown.
    ("only the spawned hosts are available", "309-311",
     (60 * MIN, 0, 0, T30), [], 3, 0, 9, 0, True, (20 * MIN, MAX, 3, False, 0, 0)),
    # named groups: over 1, durOver 1h, dur 2h + 1h = 3h, free 1, required 2 + 1 = 3 (:268-277; the "" row is skipped).
    # correctedExpected = 10h - 3h = 7h (:280); correctedOver = 3h - 1h = 2h (:282); scheduled = 5h (:284);
    # overNoTG = 4 - 1 = 3 (:286); correctedSpawned = 10 - 3 = 7 (:289); hostsAvail = (6 - 1) + 7 - 3 = 9 (:291);
    # noSpawns = 9 - 7 = 2; timeToEmpty = 5h / 9 = 2e12 ns; no-spawns = 5h / 2 = 9e12 ns (:313-314).
    ("task groups are taken out of every total", "268-291,313-314",
     (10 * HOUR, 3 * HOUR, 4, T30), [("", 7 * HOUR, 2 * HOUR, 3, 9, 9), ("g1", 2 * HOUR, 1 * HOUR, 1, 1, 2), ("g2", 1 * HOUR, 0, 0, 0, 1)],
     10, 6, 30, 0, True, (2 * 10**12, 9 * 10**12, 9, False, 0, 0)),
    # scheduled = 60m, 10 free hosts, none spawned: both = 6m = 3.6e11 ns; ratio = 3.6e11 / 1.8e12 = 0.2f < 0.25.
    # killable = int(float32(20) * (1 - 0.2f)): 1 - 0.2f rounds to 0.8f = 0.80000001..., 20 * 0.8f = 16.0000002 -> 16.0f ->
    # 16 (:398); target = 20 - 16 = 4 >= minimum 2 (:399-404).
    ("ratio 0.2: float32 drawdown arithmetic", "319,327,398-399",
     (60 * MIN, 0, 0, T30), [], 0, 10, 20, 2, True, (6 * MIN, 6 * MIN, 10, True, 4, 16)),
    # the same queue with drawdown not allowed (termination rule off / provider not spawnable / hourly billing, :325-331)
    ("drawdown not allowed", "325-331",
     (60 * MIN, 0, 0, T30), [], 0, 10, 20, 2, False, (6 * MIN, 6 * MIN, 10, False, 0, 0)),
    # the same with MinimumHosts = 10 > 20 - 16: the target is raised to the minimum (:402-404)
    ("the cap target never goes under MinimumHosts", "402-404",
     (60 * MIN, 0, 0, T30), [], 0, 10, 20, 10, True, (6 * MIN, 6 * MIN, 10, True, 10, 16)),
    # one host up: killable = int(1.0f * 0.8f) = 0 -> not > lowCountFloor: no drawdown job (:406-407)
    ("a single up host is never drawn down", "406-407",
     (60 * MIN, 0, 0, T30), [], 0, 10, 1, 0, True, (6 * MIN, 6 * MIN, 10, False, 0, 0)),
    # no host up at all: the drawdown branch is not entered (:327 len(upHosts) > 0)
    ("no up hosts", "327",
     (60 * MIN, 0, 0, T30), [], 0, 10, 0, 0, True, (6 * MIN, 6 * MIN, 10, False, 0, 0)),
    # group totals above the distro's (a queue info whose rows disagree): scheduled = (1h - 2h) - 0 = -1h <= 0 -> 0 (:299)
    ("negative scheduled duration counts as nothing queued", "299",
     (1 * HOUR, 0, 0, T30), [("g", 2 * HOUR, 0, 0, 0, 0)], 0, 4, 0, 0, False, (0, 0, 4, False, 0, 0)),
    # Go's integer division truncates: 7 ns / 2 hosts = 3 ns; without the spawned host 7 / 1 = 7 (:313-314).
    # ratio = 3 / 1.8e12 > 0 and < 0.25: killable = int(4.0f * (1 - 1.7e-12f)) = int(4.0f * 1.0f) = 4, target 0 (:398-399).
    ("integer division truncates", "313-314,398",
     (7, 0, 0, T30), [], 1, 1, 4, 0, True, (3, 7, 2, True, 0, 4)),
]
