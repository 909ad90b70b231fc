"""Parity bars with their measurements on record.

`within(name, measured, bar)` asserts measured <= bar and prints both, so that a `pytest -m gpu -s` log lists every measured
maximum next to the bar it is held to (the table of DESIGN.md section 5 is made from such a log).
VERDICT r2 #5: bars are twice the measured maximum (never below the quantum the quantity is printed / rounded with), and a
failure names the measured value."""


def within(name, measured, bar):
    measured = float(measured)
    print(f"\n[measured] {name}: {measured:.3e} (bar {bar:.3e})")
    assert measured <= bar, f"{name}: measured {measured:.3e} > bar {bar:.3e}"


def at_least(name, measured, bar):
    measured = float(measured)
    print(f"\n[measured] {name}: {measured:.6f} (bar >= {bar:.6f})")
    assert measured >= bar, f"{name}: measured {measured:.6f} < bar {bar:.6f}"
