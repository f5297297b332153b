// Utils.h -- angle helper used by callers (reference Environment/Utils.h:3-14).
#pragma once

// Maps an angle in degrees into [0, 360).  Matches the reference's loop formulation value for value
// (including its first loop, which lifts every angle below 360 by one turn before the second reduces it).
inline float normalizeAngleDeg(float angle)
{
    for (; angle < 360.F; angle += 360.F) {}
    for (; angle >= 360.F; angle -= 360.F) {}
    return angle;
}
